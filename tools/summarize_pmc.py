"""Turns the rocprofv3 --pmc CSVs under profiles/ (rNN_pmc_fetch_size.csv, rNN_pmc_write_size.csv,
optional rNN_pmc_sq.csv; python tools/summarize_pmc.py [rNN], default r02) into profiles/rNN_traffic.json: HBM-side bytes per launch for the kernels
bench.py and DESIGN.md quote.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE
counts 64 B per 128-B request for wide coalesced streams, so kernels whose reads are vector streams
are doubled; kernels whose reads are scalar loads (s_load_dwordx16 of triangle records) are left
as measured (width uncalibrated) and flagged.  WRITE_SIZE is taken as is."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"


def load(name):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(P, name)
    if not os.path.exists(path):
        return d
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


f, w, sq = load(f"{RND}_pmc_fetch_size.csv"), load(f"{RND}_pmc_write_size.csv"), load(f"{RND}_pmc_sq.csv")
sq2 = load(f"{RND}_pmc_sq2.csv")


def pick(d, key, counter):
    for k, v in d.items():
        if key in k and counter in v:
            return sum(v[counter]) / len(v[counter])
    return None


KERNELS = (  # (substring of the kernel name, key in the JSON, reads are vector streams)
    ("nn_sweep_bf16_kernel<4", "nn_sweep_bf16_kernel", True),
    ("nn_sweep_kernel<4", "nn_sweep_kernel", True),
    ("nn_sweep_kernel<1", "nn_sweep_kernel_culled", True),
    ("icp_pass_kernel", "icp_pass_kernel", True),
    ("icp_finish_kernel", "icp_finish_kernel", True),
    ("ray_sweep_mfma_kernel", "ray_sweep_mfma_kernel", True),
    ("ray_sweep_rpl_kernel<true>", "ray_sweep_rpl_kernel", False),
    ("ray_sweep_seg_kernel", "ray_sweep_seg_kernel", False),
    ("ray_cull_mask_kernel", "ray_cull_mask_kernel", True),
    ("rast_bounds_kernel", "rast_bounds_kernel", True),
    ("rast_insert_kernel", "rast_insert_kernel", True),
    ("rast_tri_kernel", "rast_tri_kernel", True),
    ("rast_item_kernel", "rast_item_kernel", True),
    ("rast_full_kernel", "rast_full_kernel", True),
    ("ray_finalize_kernel", "ray_finalize_kernel", True),
    ("erode_walk_kernel<2>", "erode_walk_kernel", True),
    ("erode_kernel<2", "erode_kernel", True),
    ("bilateral_kernel<2", "bilateral_kernel", True),
    ("xyzmap_kernel", "xyzmap_kernel", True),
)
out = {}
for key, name, stream in KERNELS:
    fe, wr = pick(f, key, "FETCH_SIZE"), pick(w, key, "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    fe, wr = fe * 1024.0, wr * 1024.0  # the counters are in KiB
    rec = {"fetch_bytes_raw": fe, "fetch_bytes_corrected": fe * (2 if stream else 1), "write_bytes": wr,
           "hbm_bytes_per_launch": fe * (2 if stream else 1) + wr,
           "note": "FETCH_SIZE doubled (coalesced vector streams, gfx950 tallies 128-B requests at 64 B)" if stream else
                   "FETCH_SIZE as measured: reads are scalar loads (s_load_dwordx16), width uncalibrated on gfx950"}
    for c in ("SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_ACTIVE_INST_VALU",
              "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
        v = pick(sq, key, c)
        if v is not None:
            rec[c] = v
    if rec.get("SQ_VALU_MFMA_BUSY_CYCLES") and rec.get("GRBM_GUI_ACTIVE"):
        # rocprofv3's MfmaUtil: busy cycles summed over the 1024 SIMDs / (kernel cycles x 1024);
        # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs
        rec["mfma_util"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        rec["mfma_flop"] = rec["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
        rec["mfma_busy"] = rec["mfma_util"]
    for c in ("SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_MFMA", "SQ_WAIT_INST_ANY", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"):
        v = pick(sq2, key, c)
        if v is not None:
            rec[c] = v
    if rec.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        rec["mfma_flop_bf16"] = rec["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
    out[name] = rec
# the triangle-driven ray stage as a whole: its kernels of one cast summed (ray_finalize_kernel is shared with the other
# variants: its mean over all casts is taken)
parts = [out[k] for k in ("rast_bounds_kernel", "rast_insert_kernel", "rast_tri_kernel", "rast_item_kernel", "rast_full_kernel",
                          "ray_finalize_kernel") if k in out]
if parts:
    agg = {"hbm_bytes_per_cast": sum(p["hbm_bytes_per_launch"] for p in parts),
           "fetch_bytes_corrected": sum(p["fetch_bytes_corrected"] for p in parts), "write_bytes": sum(p["write_bytes"] for p in parts)}
    for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"):
        if all(c in p for p in parts):
            agg[c] = sum(p[c] for p in parts)
    if agg.get("SQ_WAVE_CYCLES"):
        agg["wait_share_of_wave_cycles"] = agg["SQ_WAIT_ANY"] / agg["SQ_WAVE_CYCLES"]
    agg["per_kernel_wait_share"] = {k: out[k]["SQ_WAIT_ANY"] / out[k]["SQ_WAVE_CYCLES"] for k in out
                                    if k.startswith("rast_") and out[k].get("SQ_WAVE_CYCLES")}
    out["ray_stage_rast"] = agg
# the same stage against a resident ray set: only the triangle kernels and the result kernel run per cast
parts = [out[k] for k in ("rast_tri_kernel", "rast_item_kernel", "rast_full_kernel", "ray_finalize_kernel") if k in out]
if parts:
    out["ray_stage_rayset"] = {"hbm_bytes_per_cast": sum(p["hbm_bytes_per_launch"] for p in parts),
                               "fetch_bytes_corrected": sum(p["fetch_bytes_corrected"] for p in parts),
                               "write_bytes": sum(p["write_bytes"] for p in parts), "launches_per_cast": 4,
                               "note": "rast_tri + rast_item + rast_full + ray_finalize (the per-launch means of the PMC passes, which "
                                       "hold per-call and ray-set casts of the same frame: the same work in these four kernels)"}
json.dump(out, open(os.path.join(P, f"{RND}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
