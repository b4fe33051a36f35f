#!/usr/bin/env python3
"""Derives profiles/<name>_hbm_traffic.json from the output of tools/pmc_scene.sh.

usage: tools/make_traffic_json.py <gpurun_out/tag> <out.json> <source name written into the file>
The workload (scene, image, samples per launch) is read from the bench line of the kernel-trace
run; the render kernel is the most frequent render_mega* kernel of the PMC summary.
FETCH_SIZE / WRITE_SIZE are in KB per dispatch; FETCH_SIZE is doubled (gfx950 tallies the 128-byte
requests of wide reads at 64 bytes), WRITE_SIZE is taken as it is.
"""
import json, re, sys

tag, out, source = sys.argv[1:4]
bench = json.loads(open(f"{tag}/bench_serial.json").read().strip().splitlines()[-1])
cfg, roof = bench["config"], bench["roofline"]
kernels, cur = {}, None
for line in open(f"{tag}/pmc_summary.txt"):
    if not line.startswith(" "):
        cur = line.strip()
        kernels.setdefault(cur, {})
        continue
    m = re.match(r"\s+(\w+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
    if m and cur:
        kernels[cur][m.group(1)] = (float(m.group(2)), int(m.group(3)))
name = max((k for k in kernels if "render_mega" in k), key=lambda k: kernels[k].get("SQ_INSTS_VALU", (0, 0))[1])
c = {k: v[0] for k, v in kernels[name].items()}
spp_launch = cfg["spp_per_gpu_per_step"] / roof["launches_per_step"]
paths = cfg["width"] * cfg["height"] * spp_launch
rays = roof["counts_per_step"]["rays"] / roof["launches_per_step"]
doc = {
    "workload": {"scene": cfg["scene"], "width": cfg["width"], "height": cfg["height"], "spp_per_launch": spp_launch},
    "kernel": name,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `bench.py --steps 3 --warmup 1 --serial` "
              "(tools/pmc_scene.sh); counters are in KB per dispatch; FETCH_SIZE doubled (gfx950 reports half the bytes "
              "of wide reads), WRITE_SIZE taken as is",
    "fetch_kb_raw": c["FETCH_SIZE"], "write_kb_raw": c["WRITE_SIZE"],
    "traffic_bytes_per_launch": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
    "algorithmic_bytes_per_launch": int(paths * 16),
    "note": "algorithmic bytes: the 16-byte per-path radiance record",
    "pmc": {
        "source": source,
        "valu_wave_insts_per_launch": int(c["SQ_INSTS_VALU"]),
        "valu_wave_insts_per_ray": c["SQ_INSTS_VALU"] / rays,
        "salu_insts_per_launch": int(c["SQ_INSTS_SALU"]),
        "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4 / c["SQ_BUSY_CYCLES"] / 32,
        "active_lane_frac": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64),
        "lds_bank_conflict_frac": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1),
        "vmem_read_insts_per_launch": int(c["SQ_INSTS_VMEM_RD"]),
        "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU x 4 cycles / SQ_BUSY_CYCLES / 32 SIMDs per shader engine (instructions "
                "that issue in more than 4 cycles push it above 1); active_lane_frac = SQ_THREAD_CYCLES_VALU / "
                "(SQ_ACTIVE_INST_VALU x 64); lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
    },
}
if "TCC_HIT_sum" in c:  # the cache pass (tools/pmc_scene.sh pmc3)
    doc["pmc"]["caches"] = {
        "tcp_total_cache_accesses_per_launch": int(c["TCP_TOTAL_CACHE_ACCESSES_sum"]),
        "tcp_tcc_read_req_per_launch": int(c["TCP_TCC_READ_REQ_sum"]),
        "tcc_hit_per_launch": int(c["TCC_HIT_sum"]), "tcc_miss_per_launch": int(c["TCC_MISS_sum"]),
        "tcc_ea_rdreq_per_launch": int(c["TCC_EA_RDREQ_sum"]) if "TCC_EA_RDREQ_sum" in c else (int(c["TCC_EA0_RDREQ_sum"]) if "TCC_EA0_RDREQ_sum" in c else None),
        "l1_hit_frac": 1.0 - c["TCP_TCC_READ_REQ_sum"] / max(c["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0),
        "l2_hit_frac": c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0),
        "l1_accesses_per_ray": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / rays,
        "l1_accesses_per_cycle_and_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / 256.0 / (roof["kernel_ms_avg"] * 1e-3 * 2.4e9),
        "note": "vector L1 (TCP) accesses and the requests it passes on to L2 (TCC); L2 hits, misses and reads from memory "
                "(Infinity Cache / HBM), summed over all channels, per launch",
    }
doc["pmc"]["kernel_ms_alone_when_profiled"] = roof["kernel_ms_avg"]
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc["pmc"], indent=1))
