"""Prints the top rows of a rocprofv3 kernel_stats.csv with short kernel names.  Usage: python3 tools/kstats.py <csv> [n]"""
import csv, re, sys
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(sys.argv[1])))[:n]:
    name = r["Name"]
    m = re.search(r"(joint_\w+|rnnt_\w+|ctc_\w+|split_\w+|lane_gemm\w*|beam_\w+|greedy_\w+|hw_\w+|bfloat16\w+|reduce_kernel|Cijk_\w{0,12}|vectorized_elementwise_kernel)", name)
    short = (m.group(1) if m else name[:44])[:44]
    print("%-44s calls %5s avg ms %9.3f pct %s" % (short, r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
