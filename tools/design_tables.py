"""Markdown tables for DESIGN.md / README.md from a bench.py JSON line: python tools/design_tables.py profiles/r03_bench.json"""
import json
import sys
r = json.load(open(sys.argv[1]))
print("| shape | ms/step (pipelined) | Mpts/s | conv fwd (ms) | conv bwd (ms) | kernels | roofline of the dominant op |")
print("|---|---|---|---|---|---|---|")
for k in ("1to64", "3to8", "dw256"):
    v = r["layers"][k]
    rl = v["roofline"]
    extra = ""
    if "fwd_bwd" in rl:
        extra = "; fwd %.3f / bwd %.3f algorithmic, %.3f / %.3f executed" % (rl["fwd_bwd"]["fwd"]["algorithmic_frac"], rl["fwd_bwd"]["bwd"]["algorithmic_frac"],
                                                                          rl["fwd_bwd"]["fwd"]["executed_frac"], rl["fwd_bwd"]["bwd"]["executed_frac"])
    if "mfma_pipe_busy" in rl:
        extra += "; matrix pipe busy %.2f fwd / %.2f bwd" % (rl["mfma_pipe_busy"]["fwd"]["busy"], rl["mfma_pipe_busy"]["bwd"]["busy"])
    print("| %s | %.3f | %.1f | %.3f | %.3f | %s | %s: %.1f %s = **%.3f**%s |" % (
        k, v["ms_per_step"], v["value"] / 1e6, v["conv_ms"]["fwd"], v["conv_ms"]["bwd"], rl.get("conv_kernels", ""), rl["bound"],
        rl["achieved"], rl["unit"], rl["frac"], extra))
print()
print("| config | points (levels) | convs | ms/step | M points/s | launches/step | hierarchy ms | conv fwd+bwd ms (cached geometry) | CPU port (cores; sample) | GPU vs oracle |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, v in (r.get("configs") or {}).items():
    if "error" in v:
        print("| %s | error %s |" % (k, v["error"][:60]))
        continue
    c = v.get("cpu_baseline") or {}
    print("| %s | %d %s | %d | **%.2f** | %.1f | %.0f | %.2f | %.2f | %s points/s (%s; %s) | %s |" % (
        k, v["points"], v["level_sizes"], v["convolutions"], v["ms_per_step"], v["value"] / 1e6, v["library_launches_per_step"],
        v["hierarchy_ms"], v["conv_fwd_bwd_ms_cached_geometry"], "{:,.0f}".format(c.get("value", 0)).replace(",", " "), c.get("cores"),
        (c.get("sample") or "").split(";")[0], c.get("gpu_vs_oracle_max_rel_err_f32_layers")))
print()
print("headline", r["value"], r["ms_per_step"], "seq", r["config"]["sequential_ms_per_step"], "strong", r.get("strong") and (r["strong"]["value"], r["strong"]["ms_per_step"]))
print("cpu", r["cpu_baseline"]["value"], r["cpu_baseline"].get("single_thread", {}).get("value"))
b = r["breakdown"]
print({k: b[k]["ms"] for k in b if "ms" in b[k]})
