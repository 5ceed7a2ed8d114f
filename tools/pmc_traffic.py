#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_profile.sh into profiles/rNN_pmc_traffic.json: HBM-side bytes
per launch of every stage of bench.py's default workload (repet.sim, 180 s, 44.1 kHz stereo).
usage: tools/pmc_traffic.py <pmc dir> <out json>

Units and corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE count KiB; on gfx950 FETCH_SIZE
reports HALF the bytes of wide (16 B per lane) coalesced streaming reads, so kernels whose reads are such streams get
k = 2; 2-8 B per lane gathers and loads are uncalibrated (k = 1). WRITE_SIZE is exact for 16 B per lane stores."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import kernel_name  # noqa: E402  (full names: `stft_reg_kernel<2>`, not "void repet::")

root, out = sys.argv[1], sys.argv[2]
# kernel name fragment -> (stage, fetch correction k, note)
KERNELS = [
    ("stft_pair_kernel", "stft", 1, "4-8 B/lane"), ("stft_kernel", "stft", 1, "4-8 B/lane"),
    ("stft_reg_kernel", "stft", 2, "16 B/lane: one float4 per lane and point"),
    ("split_f16_kernel", "similarity_gemm", 2, "16 B/lane"), ("split_f16_rows_kernel", "similarity_gemm", 2, "16 B/lane"),
    ("gram_f16_big_pipe_kernel", "similarity_gemm", 2, "16 B/lane LDS-DMA"), ("gram_f16_big_kernel", "similarity_gemm", 2, "16 B/lane LDS-DMA"), ("gram_f16_kernel", "similarity_gemm", 2, "16 B/lane"),
    ("gram_kernel", "similarity_gemm", 2, "16 B/lane"),
    ("local_maxima_wave_kernel", "local_maxima", 1, "4 B/lane record loads, a few 16-byte groups of S"), ("local_maxima_kernel", "local_maxima", 2, "16 B/lane"),
    ("segment_maxima_kernel", "local_maxima", 2, "16 B/lane"),
    ("columns_from_rows_kernel", "rank_columns", 1, "4 B/lane"), ("rank_columns_kernel", "rank_columns", 2, "16 B/lane"),
    ("rows_from_code_columns_kernel", "rank_columns", 1, "4 B/lane"),
    ("mask_sim_bits_kernel", "mask_sim_select", 1, "4 B/lane gathers of plane rows"),
    ("mask_from_codes_kernel", "mask_sim", 1, "8-16 B/lane streams beside 4-byte table reads: uncalibrated"),
    ("code_planes_from_columns_kernel", "rank_columns", 2, "16 B/lane"),
    ("mask_sim_rank_kernel", "mask_sim", 1, "4-16 B/lane gathers"), ("mask_sim_nyquist", "mask_sim", 1, "4 B/lane gathers"),
    ("mask_sim_kernel", "mask_sim", 1, "4 B/lane gathers"),
    ("istft_ola_reg_kernel", "istft_ola", 2, "16 B/lane spectrum loads"), ("istft_ola", "istft_ola", 1, "4-8 B/lane"),
    # the second level of the peak picking (inside the peak-picking stage)
    ("unit_rows_f64_wg_kernel", "local_maxima", 1, "4-16 B/lane"), ("unit_rows_f64_kernel", "local_maxima", 1, "4-16 B/lane"), ("local_maxima_lite_kernel", "local_maxima", 2, "16 B/lane"),
    ("local_maxima_exact_kernel", "local_maxima", 1, "4 B/lane"),
]
# compulsory bytes of the streaming stages at cfg 2 (N = 7 938 000, C = 2, T = 7 753, F = 1 025, K = 99.85): a
# "measured" figure below them means a wrong correction factor, and the tool refuses to write it
N_, C_, T_, F_, K_ = 7938000, 2, 7753, 1025, 99.85
COMPULSORY = {"stft": 4 * N_ * C_ + 12 * F_ * T_ * C_ + 4 * F_ * T_, "similarity_gemm": 4 * F_ * T_ + 4 * T_ * T_,
              # round 4: the peak picking reads the segment records (three planes of 244 entries per row), not S, and writes the lists
              "local_maxima": 12 * 244 * T_ + 4 * K_ * T_, "istft_ola": 8 * F_ * T_ * C_ + 4 * N_ * C_}
SURVEY_8D = {"local_maxima": 4 * T_ * T_ + 4 * K_ * T_}      # what SURVEY 8d counts for K4 (S read once): kept beside the measured bytes
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
stages = {}
for kernel, counters in acc.items():
    hit = next((k for k in KERNELS if k[0] in kernel), None)
    if hit is None:
        continue
    _, stage, k, width = hit
    fetch = sum(counters.get("FETCH_SIZE", [0.0])) / max(len(counters.get("FETCH_SIZE", [0.0])), 1) * 1024.0
    write = sum(counters.get("WRITE_SIZE", [0.0])) / max(len(counters.get("WRITE_SIZE", [0.0])), 1) * 1024.0
    st = stages.setdefault(stage, {"hbm_bytes_per_launch": 0.0, "fetch_bytes": 0.0, "write_bytes": 0.0, "kernels": [], "fetch_correction": ""})
    st["fetch_bytes"] += fetch * k
    st["write_bytes"] += write
    st["hbm_bytes_per_launch"] += fetch * k + write
    st["kernels"].append({"kernel": kernel_name(kernel), "fetch_counted": fetch, "k": k, "write": write})
    st["fetch_correction"] = (st["fetch_correction"] + "; " if st["fetch_correction"] else "") + f"x{k} ({width})"
doc = {"_about": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_profile.sh) for bench.py's default workload "
                 "(repet.sim, 180 s, 44.1 kHz stereo), mean per launch. Bytes = WRITE_SIZE*1024 + FETCH_SIZE*1024*k, k = 2 for kernels whose "
                 "reads are 16-byte-per-lane streams (gfx950 counts those at one half), k = 1 (uncalibrated) for narrower loads and gathers. "
                 "Infinity-Cache hits are included in FETCH_SIZE (MI355X_MICROARCH.md), so this is an upper bound on HBM traffic.",
       "library_sha256": __import__("hashlib").sha256(open(__import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))),
                                                                                   "repet-python_amd", "lib", "librepet_hip.so"), "rb").read()).hexdigest(),
       "stages": stages}
for stage, need in COMPULSORY.items():
    if stage in stages:
        stages[stage]["compulsory_bytes"] = need
        stages[stage]["measured_over_compulsory"] = round(stages[stage]["hbm_bytes_per_launch"] / need, 3)
        assert stages[stage]["hbm_bytes_per_launch"] >= 0.97 * need, (stage, stages[stage]["hbm_bytes_per_launch"], need, "below the bytes the stage must move: wrong fetch correction?")
for stage, b8d in SURVEY_8D.items():
    if stage in stages:
        stages[stage]["survey_8d_bytes"] = b8d
        stages[stage]["measured_over_survey_8d"] = round(stages[stage]["hbm_bytes_per_launch"] / b8d, 3)
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in stages.items()}))
