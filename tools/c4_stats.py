"""Prints the rpf kernels of gpurun_out/c4/c4_kernel_stats.csv (tools/gpu_profile_c4.sh)."""
import csv, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in csv.DictReader(open(os.path.join(root, "gpurun_out", "c4", "c4_kernel_stats.csv"))):
    n = r["Name"]
    if "rpf" not in n:
        continue
    short = n[n.find("namespace)::") + 12:][:90].replace("rpf::(anonymous namespace)::", "")
    print("%-92s calls %5s avg %10.1f us" % (short, r["Calls"], float(r["AverageNs"]) / 1e3))
