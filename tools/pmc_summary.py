"""Turns the rocprofv3 counter CSVs collected by tools/gpu_pmc.sh (gpurun_out/pmc/) into the committed summaries
profiles/rNN_pmc_traffic.json (HBM bytes per launch, with the FETCH_SIZE/WRITE_SIZE calibration) and rNN_pmc_sq.json."""
import collections, csv, json, sys

P = "gpurun_out/pmc/"
RND = sys.argv[1] if len(sys.argv) > 1 else "r01"


def mean(path, kern, ctr):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Kernel_Name"].startswith(kern) and r["Counter_Name"] == ctr]
    return sum(v) / len(v), len(v)


GiB_KiB = 1048576.0
cal = {}
for k, kr, kw in (("copy16(", GiB_KiB, GiB_KiB), ("copy16_nt(", GiB_KiB, GiB_KiB), ("copy4_12(", (2**30 // 12) * 12 / 1024, (2**30 // 12) * 12 / 1024),
                  ("mix_1r2w(", GiB_KiB, 2 * GiB_KiB), ("mix_1r2w_strided(", GiB_KiB, 2 * GiB_KiB)):
    f, _ = mean(P + "calib_FETCH_SIZE/c_counter_collection.csv", k, "FETCH_SIZE")
    w, _ = mean(P + "calib_WRITE_SIZE/c_counter_collection.csv", k, "WRITE_SIZE")
    cal[k.rstrip("(")] = {"known_read_KiB": kr, "FETCH_SIZE_KiB": f, "fetch_ratio": round(f / kr, 5), "known_write_KiB": kw, "WRITE_SIZE_KiB": w,
                          "write_ratio": round(w / kw, 5)}
names = set(r["Kernel_Name"] for r in csv.DictReader(open(P + "bench_FETCH_SIZE/b_counter_collection.csv")) if "k_nv12_rgb_p16" in r["Kernel_Name"])
kern = sorted(names)[0].split("(")[0]
f, nf = mean(P + "bench_FETCH_SIZE/b_counter_collection.csv", kern, "FETCH_SIZE")
w, nw = mean(P + "bench_WRITE_SIZE/b_counter_collection.csv", kern, "WRITE_SIZE")
frames, alg = 32, 12441600 + 24883200
traffic = (f * 2.0 + w) * 1024
out = {"round": RND, "kernel": kern, "frames_per_launch": frames, "counter_unit": "KiB", "FETCH_SIZE_per_launch": f, "WRITE_SIZE_per_launch": w,
       "n_dispatches_fetch": nf, "n_dispatches_write": nw,
       "corrections": {"FETCH_SIZE": "x2.0 (gfx950 tallies 128-B read requests at 64 B; measured %.5f on copy16 with a known 1 GiB read, same 16 B/lane loads as this kernel)" % cal["copy16"]["fetch_ratio"],
                       "WRITE_SIZE": "x1.0 (measured %.4f on copy16, %.4f on copy16_nt, %.4f on copy4_12)" % (cal["copy16"]["write_ratio"], cal["copy16_nt"]["write_ratio"], cal["copy4_12"]["write_ratio"])},
       "hbm_read_bytes_per_launch": f * 2.0 * 1024, "hbm_write_bytes_per_launch": w * 1024, "hbm_bytes_per_launch": traffic, "hbm_bytes_per_frame": traffic / frames,
       "algorithmic_bytes_per_frame": alg, "traffic_over_algorithmic": traffic / frames / alg, "calibration": cal,
       "how": "tools/gpu_pmc.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only) over `bench.py --steps 6 --warmup 2 --no-cpu` and tools/pmc_calib.bin"}
json.dump(out, open(f"profiles/{RND}_pmc_traffic.json", "w"), indent=1)
print(kern, "traffic/algorithmic =", round(out["traffic_over_algorithmic"], 5))

sq = collections.defaultdict(list)
for d in ("bench_SQ", "bench_TCC"):
    try:
        for r in csv.DictReader(open(P + d + "/b_counter_collection.csv")):
            if r["Kernel_Name"].startswith(kern):
                sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
    except FileNotFoundError:
        pass
if sq:
    m = {k: sum(v) / len(v) for k, v in sq.items()}
    waves = m.get("SQ_WAVES", 0)
    s = {"round": RND, "kernel": kern, "per_launch_of_32_frames": m,
         "derived": {"waves": waves, "valu_wave_instructions_per_wave": m.get("SQ_INSTS_VALU", 0) / waves if waves else None,
                     "valu_ops_per_pixel": (m.get("SQ_INSTS_VALU", 0) * 64) / (32 * 3840 * 2160 * (4096 / 3840)) if waves else None,
                     "lds_instructions_per_wave": m.get("SQ_INSTS_LDS", 0) / waves if waves else None,
                     "lds_bank_conflict_cycles": m.get("SQ_LDS_BANK_CONFLICT"),
                     "l2_miss_bytes_if_128B": m.get("TCC_MISS_sum", 0) * 128}}
    json.dump(s, open(f"profiles/{RND}_pmc_sq.json", "w"), indent=1)
    print("LDS bank conflict cycles:", m.get("SQ_LDS_BANK_CONFLICT"), " VALU wave-instr per wave:", s["derived"]["valu_wave_instructions_per_wave"])
