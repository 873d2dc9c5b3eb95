#!/usr/bin/env python3
"""Turn a pmc_summary.py CSV into profiles/<round>_pmc_kernels.json: the HBM bytes per launch of the kernels bench.py's
`roofline` object names (gate/up GEMM + SiLU epilogue = k_gemm<1, false, 1, true>, the largest share of the cycle; lm_head GEMM
+ fused argmax = k_gemm<1, false, 2, true>), stamped with the hash of the sources the kernels are built from so that bench.py
can refuse the numbers once a kernel has changed.
usage: pmc_kernels_json.py <summary.csv> <out.json> [<kernel_stats.csv of the rocprofv3 --kernel-trace --stats run>]
(the third argument adds each kernel's average duration by rocprofv3, which bench.py prints beside its own HIP-event figure)"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["dflash_amd/csrc/gemm_skinny.hip", "dflash_amd/csrc/gemm_rows.h", "dflash_amd/csrc/dfl_common.h"]
# --batch: the 4-request leg (bench.py --requests-per-gpu 4): the ragged-batch lm_head / gate-up kernels of gemm_batch.hip
BATCH_SOURCES = ["dflash_amd/csrc/gemm_batch.hip", "dflash_amd/csrc/gemm_ring.h", "dflash_amd/csrc/gemm_rows.h", "dflash_amd/csrc/dfl_common.h"]
# (round 4: the ring form, gemm_ring.h: dfl_k_gemm_r<MT, TPU, KQ, NW, A, EPI, CK>)
BATCH_KERNELS = {"lm_head": ("dfl_k_gemm_r<4,1,1,16,2,2,8>", 151936 * 4096 * 2), "gate_up": ("dfl_k_gemm_r<4,2,4,12,2,1,8>", 2 * 12288 * 4096 * 2)}
# (round 3 names: k_gemm<MT, CHUNKED, EPI, NORM>; the normalised-source instantiations are the ones the cycle runs)
KERNELS = {"gate_up": ("k_gemm<1,false,1,true>", 2 * 12288 * 4096 * 2), "lm_head": ("k_gemm<1,false,2,true>", 151936 * 4096 * 2)}


def source_hash(sources=None):
    h = hashlib.sha256()
    for f in (sources or KERNEL_SOURCES):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    if "--batch" in sys.argv:
        sys.argv.remove("--batch")
        KERNEL_SOURCES, KERNELS = BATCH_SOURCES, BATCH_KERNELS
    allrows = list(csv.DictReader(open(sys.argv[1])))
    out = {"kernels": {}, "source": os.path.relpath(sys.argv[1], ROOT), "kernel_source_sha256_16": source_hash(KERNEL_SOURCES),
           "kernel_sources": KERNEL_SOURCES,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/profile_gpu.sh pmc); "
                     "hbm = 2 * FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of a wide coalesced read, "
                     "MI355X_MICROARCH.md, HBM)"}
    for key, (kname, alg) in KERNELS.items():
        rows = [r for r in allrows if r["kernel"].replace(" ", "") == kname]
        if not rows:
            continue
        r = max(rows, key=lambda x: int(x["dispatches"]))
        out["kernels"][key] = {"kernel": r["kernel"], "fetch_size_kb": float(r["FETCH_SIZE_KB_mean"]),
                               "write_size_kb": float(r["WRITE_SIZE_KB_mean"]),
                               "hbm_bytes_per_launch": int(r["hbm_bytes_corrected"]), "algorithmic_bytes": alg,
                               "traffic_over_algorithmic": int(r["hbm_bytes_corrected"]) / alg,
                               "dispatches": int(r["dispatches"])}
    if len(sys.argv) > 3:   # rocprofv3 --stats: "Name","Calls","TotalDurationNs","AverageNs",...
        out["kernel_stats_source"] = os.path.relpath(sys.argv[3], ROOT)
        for r in csv.DictReader(open(sys.argv[3])):
            nm = r["Name"].replace(" ", "")
            for key, (kname, _) in KERNELS.items():
                if ("::" + kname + "(" in nm or nm.startswith("void" + kname + "(") or nm.startswith(kname + "(")) and key in out["kernels"]:
                    out["kernels"][key]["rocprof_avg_us"] = float(r["AverageNs"]) / 1e3
                    out["kernels"][key]["rocprof_calls"] = int(r["Calls"])
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out))
