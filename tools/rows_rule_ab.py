"""The executor's kernel-family rule for depth-wise layers (csrc/exec.hip rows_shape) against row kernels everywhere
(MCCNN_DEBUG=rows_force=1), per layer of a configuration with geometry and plans cached, and the pipelined step:
    python tools/rows_rule_ab.py cfg3      (run once per setting: the switch is read once per process)"""
import os, sys, types, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_device(0)
torch.autograd.set_multithreading_enabled(False)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
e = bench.run_config(name, torch.device("cuda", 0), types.SimpleNamespace(no_pipeline=False), False)
print("MCCNN_DEBUG=%s %s: %s %.3f ms/step" % (os.environ.get("MCCNN_DEBUG", ""), name, e["mode"], e["ms_per_step"]))
for l in e["layers"]:
    print("   %-9s F=%-4d pts %-7d centres %-7d e %-8d fwd %.4f bwd %.4f" % (l["name"], l["fin"], l["points_in"], l["centres"], l["edges"], l["fwd_ms"], l["bwd_ms"]))
