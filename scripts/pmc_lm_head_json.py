#!/usr/bin/env python3
"""Turn a pmc_summary.py CSV into profiles/<round>_pmc_lm_head.json: the HBM bytes per launch of the dominant
kernel (lm_head GEMM + fused argmax, k_gemm<1, false, 2>), stamped with the hash of the sources the kernel is
built from so that bench.py can refuse the number once the kernel has changed.
usage: pmc_lm_head_json.py <summary.csv> <out.json>"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["dflash_amd/csrc/gemm_skinny.hip", "dflash_amd/csrc/gemm_rows.h", "dflash_amd/csrc/dfl_common.h"]


def source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["kernel"].replace(" ", "") == "k_gemm<1,false,2>"]
    r = max(rows, key=lambda x: int(x["dispatches"]))
    out = {"kernel": "k_gemm<1, false, 2>", "fetch_size_kb": float(r["FETCH_SIZE_KB_mean"]),
           "write_size_kb": float(r["WRITE_SIZE_KB_mean"]), "hbm_bytes_per_launch": int(r["hbm_bytes_corrected"]),
           "algorithmic_bytes": 151936 * 4096 * 2, "dispatches": int(r["dispatches"]),
           "source": os.path.relpath(sys.argv[1], ROOT), "kernel_source_sha256_16": source_hash(),
           "kernel_sources": KERNEL_SOURCES,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (scripts/profile_gpu.sh pmc); "
                     "hbm = 2 * FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of a wide coalesced read, "
                     "MI355X_MICROARCH.md, HBM)"}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out))
