"""Mean FETCH_SIZE / WRITE_SIZE (KB) per launch of the char-conv GEMM (gate epilogue, big tiles) from the
two rocprofv3 --pmc CSVs -> profiles/<name>_traffic.json.
    python scripts/pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv out.json [CfgI8]
(the optional 4th argument names the tile configuration of the kernel: CfgBig = bf16x3 tiles, the
default; CfgI8 = the int8x2 tile)"""
import csv
import json
import re
import sys


TAG = sys.argv[4] if len(sys.argv) > 4 else "CfgBig"


def mean_kb(path, counter):
    """mean over the launches of the bench batch: the first four distinct grids of the kernel in dispatch order are
    char layers 1-4 of the timed batch (later legs of bench.py launch other shapes)"""
    vals, grids = {}, []
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if "gemm_kernel<4" in n and TAG in n and r["Counter_Name"] == counter:
                if TAG == "CfgI8" and not re.search(r"CfgI8, true, false>", n):
                    continue                    # the word convolution's instance of the tile
                if r["Grid_Size"] not in grids and len(grids) < 4:
                    grids.append(r["Grid_Size"])
                if r["Grid_Size"] in grids:
                    vals.setdefault(r["Grid_Size"], []).append(float(r["Counter_Value"]))
    per_layer = [sum(v) / len(v) for v in vals.values()]          # one mean per layer, then the mean of the layers
    return sum(per_layer) / max(len(per_layer), 1), sum(len(v) for v in vals.values())


fetch, nf = mean_kb(sys.argv[1], "FETCH_SIZE")
write, nw = mean_kb(sys.argv[2], "WRITE_SIZE")
out = {
    "kernel": "gemm_kernel<GATE, %s> (char conv, layers>=1)" % ("int8x2, CfgI8" if TAG == "CfgI8" else "bf16x3, big tiles"),
    "config": 1,
    "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write, "launches": [nf, nw],
    "correction": "gfx950: FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane loads (LDS-DMA) at 64 bytes: "
                  "doubled; WRITE_SIZE exact (MI355X_MICROARCH.md, HBM)",
    "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
    "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `bench.py --steps 5 --warmup 2` "
           "(scripts/pmc_pass.sh, serial run), mean over the launches of char layers 1-4",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
