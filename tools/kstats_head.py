import pandas as pd, sys
d=pd.read_csv(sys.argv[1])
d["n"]=d.Name.str.replace("void mccnn::","").str.replace("mccnn::","").str.slice(0,44)
print(d[["n","Calls","AverageNs"]].head(18).to_string())
