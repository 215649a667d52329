import csv, glob, os, sys
root = sys.argv[1]
order = ["1", "2", "3", "11", "12", "10", "13", "4", "5", "6", "7", "8", "full"]
names = {"1": "load", "2": "moves", "3": "consume + n-shuffle + beams", "11": "spawn: thresholds + bulk rng (+twist)", "12": "spawn: apple scan",
         "10": "spawn: t* + shuffle draws", "13": "spawn: shuffle apply", "4": "spawn: waste pick + writes", "5": "rewards",
         "6": "features + contract + metric loads", "7": "metric update/stores + done", "8": "state stores", "full": "obs"}
def load(k):
    f = sorted(glob.glob("%s/%s/**/*counter_collection.csv" % (root, k), recursive=True), key=os.path.getmtime)
    acc = {}
    for row in csv.DictReader(open(f[-1])):  # the newest run (gpurun merges every run's files into the same directory)
        if "k_grid_step" in row["Kernel_Name"]:
            s, c = acc.get(row["Counter_Name"], (0.0, 0)); acc[row["Counter_Name"]] = (s + float(row["Counter_Value"]), c + 1)
    w = acc["SQ_WAVES"][0] / acc["SQ_WAVES"][1]
    return {c: s / k2 / w for c, (s, k2) in acc.items()}
prev = {}
cols = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SMEM"]
short = ["VALU", "SALU", "LDS", "LDS_active_cyc", "LDS_conflict", "SMEM"]
print("# k_grid_step<cleanup>, n = 8, 16384 envs, steady state: executed instructions PER ENV-STEP (= per wave), by phase.")
print("# Method: the kernel is built 12 times, each build ending at one more phase boundary (-DCE_TRUNCATE=k, everything live")
print("# folded into one store); the PMC counters of consecutive builds are differenced.  The compiler schedules each build")
print("# on its own, so a phase that mostly hoists work of the NEXT phase can come out a few instructions negative: read")
print("# the cumulative column for anything finer than ~10 instructions.  Never shipped, never benchmarked.")
print("%-42s" % "phase" + "".join("%14s" % c for c in short) + "%14s" % "cum VALU" + "%10s" % "cum SALU")
for k in order:
    cur = load(k)
    print("%-42s" % names[k] + "".join("%14.1f" % (cur.get(c, 0) - prev.get(c, 0)) for c in cols)
          + "%14.1f%10.1f" % (cur.get("SQ_INSTS_VALU", 0), cur.get("SQ_INSTS_SALU", 0)))
    prev = cur
print("%-42s" % "total (the shipped kernel)" + "".join("%14.1f" % prev.get(c, 0) for c in cols))
