"""Regenerate the round-2 section of profiles/README.md from the CSV summaries under profiles/r02_*: python tools/profiles_readme.py"""
import csv, collections, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
def sq(tag):
    agg = collections.defaultdict(dict); meta = {}
    for r in csv.DictReader(open(f"profiles/{tag}/pmc_sq.csv")):
        agg[r["kernel"]][r["counter"]] = float(r["mean_per_dispatch"]); meta[r["kernel"]] = (r["vgpr"], r["agpr"], r["scratch"])
    out = []
    for k, m in agg.items():
        w = m["SQ_WAVES"]; wc = m["SQ_WAVE_CYCLES"] * 4
        out.append((k.replace("void klhr::", ""), f"{m['SQ_ACTIVE_INST_VALU']*4/wc*100:.0f} %", f"{m['SQ_WAIT_ANY']*4/wc*100:.0f} %",
                    f"{m['SQ_THREAD_CYCLES_VALU']/(m['SQ_ACTIVE_INST_VALU']*64)*100:.0f} %", f"{wc/m['SQ_INSTS_VALU']:.1f}", f"{m['SQ_INSTS_VALU']/w:.3g}", "/".join(meta[k])))
    return out
def ks(tag):
    rows = [r for r in csv.DictReader(open(f"profiles/{tag}/kernel_stats.csv")) if "klhr" in r["Name"] and "init" not in r["Name"]]
    return [(r["Name"].split("(")[0].replace("void klhr::", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["Percentage"])) for r in rows]
DESC = {"r02_octet_ill": "ill-normal-100 × 16 384, round-1 octet kernel (baseline of round 2)", "r02_octet_f100": "funnel-100 × 131 072, round-1 octet kernel (baseline)",
        "r02_octet_glmm": "glmm-poisson-45 × 8 192, round-1 octet kernel (baseline)", "r02_octet_glmm_now": "glmm-poisson-45 × 8 192, staged model data read as LDS",
        "r02_octet_glmm_final": "glmm-poisson-45 × 8 192, end of round 2 (`exp` over year blocks, two-loop fit)",
        "r02_split_ill": "ill-normal-100 × 16 384, split pipeline, first form (one wavefront per tile, lane = chain fit kernels)",
        "r02_split_ill_final": "ill-normal-100 × 16 384, end of round 2 (pair-mode streaming kernel, node-parallel fit)",
        "r02_split_f100": "funnel-100 × 131 072, split pipeline, end of round 2 (both fit stages in one launch)",
        "r02_lane_f10": "funnel-10 × 65 536 (headline), lane kernel with closed-form fits, end of round 2"}
txt = "\n\n## Round 2 (`r02_*`): one directory per configuration and kernel mapping\n\n"
txt += ("Commands: `tools/profile_cfg.sh <tag> <model> <dim> <chains> [steps]` on the GPU box (kernel-trace `--stats`, two SQ counter passes, `--pmc FETCH_SIZE`, "
        "`--pmc WRITE_SIZE`, each its own run of `bench.py --model … --no-configs`; the split pipeline with `KLHR_GROUPS=1` so that kernels do not overlap), summaries copied by "
        "`tools/profile_cfg_collect.py`, which also writes `traffic.json[<model>-<dim>-<chains>]` (what `bench.py` reports as `roofline.traffic`; FETCH_SIZE ÷ 0.5 for the tiled reads "
        "of the split pipeline, ÷ 1.383 for the lane-private rows of the monolithic kernels: `r02_calibration/rowcal.csv`).  Each directory: `kernel_stats.csv` (rocprofv3 `--stats`), "
        "`pmc_sq.csv` (mean per dispatch over the last third of the dispatches), `pmc_fetch_write.csv`.  The average durations include the 150 untimed setup draws (window closures, first "
        "launches); `bench.py`'s `roofline.kernels` are HIP-event means over 20 post-setup iterations and sit between the minimum and the average here.  (This section is generated: "
        "`tools/profiles_readme.py`.)\n\n")
txt += "| dir | configuration, build | kernels: calls, average / minimum duration (µs), share |\n|---|---|---|\n"
for tag, d in DESC.items():
    txt += f"| `{tag}/` | {d} | " + "; ".join(f"`{n}` ×{c}: {a:.1f} / {m:.1f}, {p:.1f} %" for n, c, a, m, p in ks(tag)) + " |\n"
txt += ("\nSQ counters (per kernel: VALU-active share of the wave cycles, share waiting in `s_waitcnt`, lanes active on issued VALU instructions, cycles per VALU instruction, VALU "
        "instructions per wavefront and dispatch, VGPR/AGPR/scratch as rocprofv3 reports them):\n\n| dir | kernel | VALU-active | waiting | lanes active | cycles / VALU | VALU / wavefront | VGPR/AGPR/scratch |\n|---|---|---|---|---|---|---|---|\n")
for tag in DESC:
    for row in sq(tag):
        txt += f"| `{tag}/` | `{row[0]}` | {row[1]} | {row[2]} | {row[3]} | {row[4]} | {row[5]} | {row[6]} |\n"
s = open("profiles/README.md").read()
i = s.index("\n\n## Round 2 (`r02_*`)"); j = s.index("\n`r02_stamps/`")
s = s[:i] + txt + s[j:]

# ---- round 3: the four BASELINE configurations on the final build of the round, every traffic.json entry tied to its build id
import json
R3 = {"r03_lane_f10": "funnel-10 × 65 536 (headline), lane kernel", "r03_split_f100": "funnel-100 × 131 072, split pipeline (direction cache), `KLHR_GROUPS=1`",
      "r03_tile_ill": "ill-normal-100 × 16 384, tile kernel (whole iterations per 16-chain tile: quad-mode streaming body + node-parallel fit body, 50 draws per launch)",
      "r03_octet_glmm": "glmm-poisson-45 × 8 192, octet kernel"}
tj = json.load(open("profiles/traffic.json"))
t3 = "\n\n## Round 3 (`r03_*`): the four BASELINE configurations on the round's final build\n\n"
t3 += ("Same commands as round 2 (`tools/profile_cfg.sh`, python directly after `rocprofv3 … --`).  New: every `traffic.json` entry carries `build_id` = `klhr_build_id()` "
       "(sha256 of the library's sources) of the library it was measured on, and `bench.py` quotes `roofline.traffic` / `fp64_valu.measured` only while the loaded library is that "
       "build (`traffic_stale: true` + reason otherwise); the entry's kernel is the one that moves the iteration's algorithmic bytes (streaming kernel / hot kernel), also when a fit "
       "kernel is longer.  FETCH_SIZE ÷ 0.5 (tiled reads) or ÷ 1.40 (lane-private rows: mean of three calibration runs, 1.383 / 1.455 / 1.357, `r01b` / `r02_calibration` / "
       "`r03_calibration`, which now also holds the `tile_write` WRITE_SIZE rows: 1.0000).  NOTE on register counts: rocprofv3's `VGPR_Count` column is (architectural + accumulation "
       "registers) / 2 on this stack — the lane kernel's \"256\" is 256 + 256, `klhr_fit12_kernel`'s \"140\" is 256 + 18 (one wavefront per SIMD, as the compiler's resource remarks say); "
       "round 2's DESIGN text read these columns as VGPR counts.  (This section is generated: `tools/profiles_readme.py`.)\n\n")
t3 += "| dir | configuration | kernels: calls, average / minimum duration (µs), share |\n|---|---|---|\n"
for tag, d in R3.items():
    t3 += f"| `{tag}/` | {d} | " + "; ".join(f"`{n}` ×{c}: {a:.1f} / {m:.1f}, {p:.1f} %" for n, c, a, m, p in ks(tag)) + " |\n"
t3 += ("\n| dir | kernel | VALU-active | waiting | lanes active | cycles / VALU | VALU / wavefront | (VGPR+AGPR)/2 / - / scratch |\n|---|---|---|---|---|---|---|---|\n")
for tag in R3:
    for row in sq(tag):
        t3 += f"| `{tag}/` | `{row[0]}` | {row[1]} | {row[2]} | {row[3]} | {row[4]} | {row[5]} | {row[6]} |\n"
t3 += "\n| `traffic.json` key | kernel | HBM bytes / launch (FETCH corrected + WRITE) | algorithmic bytes / launch | ratio | counter-derived fp64 GFLOP / launch | rocprof ms per iteration vs bench |\n|---|---|---|---|---|---|---|\n"
for k, v in tj.items():
    if "build_id" in v:
        t3 += (f"| `{k}` | `{v['kernel'].replace('void klhr::', '')}` | {v['hbm_bytes_per_launch']/1e9:.3f} GB | {v['algorithmic_bytes_per_launch']/1e9:.3f} GB | "
               f"{v['hbm_bytes_per_launch']/v['algorithmic_bytes_per_launch']:.2f} | {v['fp64_flops_per_launch']/1e9:.1f} | {v['rocprof_ms_per_iteration_all_kernels']:.4f} vs {v['bench_ms_per_step_when_profiled']:.4f} |\n")
t3 += ("\n`r03_stamps/`: `tile_kernel_cycle_split.txt` — per-phase cycles of the tile kernel's two wavefronts and of its fit (ill-normal-100 x 16 384 / 8 192, funnel-100 x 16 384) and of "
       "the single-wavefront 16-chain streaming kernel, from a `-DKLHR_STAMPS` build checked bit for bit against the production library (`tools/stamps_tile.sh`); cycle split and "
       "bench of the free-running octet kernel (built, measured slower, removed: DESIGN.md §8).  `r03_experiments/`: `t1_pca_fusion.txt` (CCIPCA's first pass fused into T1: bit-identical, "
       "0-1 %, no traffic saved: reverted); `klhr_fit12_kernel` with a pass cap "
       "(what the KL fits' tail costs) and with straggler compaction in several workgroup / register configurations, funnel-100 × 131 072, same box back to back.\n")
k3 = "\n\n## Round 3 (`r03_*`)"
if k3 in s:
    s = s[:s.index(k3)]
s = s.rstrip("\n") + t3
open("profiles/README.md", "w").write(s)
print("profiles/README.md regenerated")
