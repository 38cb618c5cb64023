"""Builds profiles/<tag>_* from the outputs of tools/final_round.sh in gpurun_out/ (run in the build container):
copies the evidence files, derives <tag>_pmc_traffic.json from the counter passes and writes <tag>_README.md.
    python tools/make_profiles.py r04 counters      after `tools/final_round.sh r04 counters` (counter files only; commit them)
    python tools/make_profiles.py r04               after `tools/final_round.sh r04 bench`"""
import csv, json, os, shutil, sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out") + "/", os.path.join(ROOT, "profiles") + "/"
KEEP_OLD = ("%s_sweep_solver_cfg.txt" % TAG, "%s_winograd_counters.txt" % TAG)


def rows(fn):
    return list(csv.DictReader(open(G + fn)))


for f in os.listdir(G):
    if f.startswith(TAG + "_") and f.endswith((".csv", ".json", ".txt")) and f != TAG + "_sweep_cfg.txt":
        shutil.copy(G + f, P + f)

# ---- counter-derived HBM traffic
passes = 13            # tools/bench_conv.py: 3 warm-up + 10 timed encoder passes
conv_kernels = ("conv3x3_x3_kernel", "stem_x3_kernel", "wino_conv_kernel", "wino_reg_kernel", "conv2d_vec_kernel", "conv2d_kernel", "conv2d_stem_kernel", "stem_conv_kernel", "conv_splitk_reduce_kernel")
fetch16 = ("wino_conv_kernel", "wino_reg_kernel", "conv2d_vec_kernel", "conv_splitk_reduce_kernel")
F = {r["kernel"]: r for r in rows(TAG + "_pmc_conv_FETCH_SIZE.csv")}
Wr = {r["kernel"]: r for r in rows(TAG + "_pmc_conv_WRITE_SIZE.csv")}
fr = fc = wr = launches = 0.0
per_kernel = {}
short = lambda k: k.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
for k, r in F.items():
    if any(c in k for c in conv_kernels):
        s = float(r["sum"]) / passes
        fr += s; fc += s * (2 if any(c in k for c in fetch16) else 1); launches += int(r["launches"]) / passes
        per_kernel[short(k)] = {"launches_per_pass": int(r["launches"]) / passes, "FETCH_KiB_per_pass": round(s, 1)}
for k, r in Wr.items():
    if any(c in k for c in conv_kernels):
        s = float(r["sum"]) / passes; wr += s
        per_kernel.setdefault(short(k), {})["WRITE_KiB_per_pass"] = round(s, 1)
r1 = json.load(open(P + "r01_pmc_traffic.json"))
# the solver's traffic: ONE source, the counter file bench.py's `counters_file` names (tools/prof_solver_counters.sh)
_sc = json.load(open(G + TAG + "_solver_counters.json")) if os.path.exists(G + TAG + "_solver_counters.json") else json.load(open(P + TAG + "_solver_counters.json"))
fk, wk = float(_sc["FETCH_SIZE"]), float(_sc["WRITE_SIZE"])
sol_f = {"launches": 6}
def index_max_entry(C, fallback):
    try:
        f = [r for r in rows("%s_pmc_index_max_C%d_FETCH_SIZE.csv" % (TAG, C)) if "index_max" in r["kernel"]][0]
        w = [r for r in rows("%s_pmc_index_max_C%d_WRITE_SIZE.csv" % (TAG, C)) if "index_max" in r["kernel"]][0]
    except (OSError, IndexError):
        return fallback, True
    fk, wk = float(f["mean_per_launch"]), float(w["mean_per_launch"])
    alg = 32 * (4 * C * 20480 + 4 * 20480 + 2 * 4 * C * 128)
    return {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_corrected": fk * 2048 + wk * 1024, "algorithmic_bytes": alg,
            "launches": int(f["launches"]), "note": "cold inputs (tools/bench_index_max.py rotates > 1 GB of buffers); FETCH x2 (16-byte loads)"}, False


im64, old64 = index_max_entry(64, r1["index_max_C64_B32_N20480_K128"])
im32, old32 = index_max_entry(32, r1["index_max_C32_B32_N20480_K128"])
out = {"units": r1["units"] + "; solver records and the 16-byte staged convolution kernels corrected x2; WRITE_SIZE raw",
       "commands": ["tools/profile_round.sh %s pmc: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python tools/bench_conv.py (13 encoder passes) | tools/bench_index_max.py; solve_kernel: copied from %s_solver_counters.json (tools/prof_solver_counters.sh, the same passes as its instruction counters)" % (TAG, TAG)],
       "carried_over_from_r01": [n for n, o in (("index_max_C64_B32_N20480_K128", old64), ("index_max_C32_B32_N20480_K128", old32)) if o],
       "index_max_C64_B32_N20480_K128": im64, "index_max_C32_B32_N20480_K128": im32,
       "solve_kernel_F32_R60_N20480": {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "fetch_bytes_corrected": fk * 2048, "hbm_bytes_corrected": fk * 2048 + wk * 1024,
                                       "compulsory_bytes": 32 * 20480 * 16 + 32 * 320 * 32, "launches": int(sol_f["launches"]),
                                       "note": "one launch = 32 frames x 60 hypotheses; a frame's records + boxes (338 KB) are read ~64 sweeps x 60 hypotheses times but stay cache resident; WRITE_SIZE = the outputs + the classification-cache entries (16 B per missed cluster and sweep; the L2 is write-through: every store leaves it) + the kernel's private segment (tools/kernel_resources.py prints its size).  Same numbers as %s_solver_counters.json: they ARE that file's" % TAG},
       "conv2d_resnet34_B32_160x512": {"FETCH_SIZE_KiB_per_encoder_pass_raw": fr, "FETCH_SIZE_KiB_per_encoder_pass_corrected": fc, "WRITE_SIZE_KiB_per_encoder_pass_raw": wr,
                                       "kernel_launches_per_pass": launches, "conv_calls_per_pass": 36, "hbm_bytes_per_call_corrected": (fc + wr) * 1024 / 36,
                                       "hbm_bytes_per_call_raw": (fr + wr) * 1024 / 36, "per_kernel": per_kernel,
                                       "note": "sum over the convolution kernels of one image-encoder pass (bf16x3 direct convolutions, any Winograd / implicit-GEMM launches left, the direct stem); FETCH x2 for the kernels that load 16 B per lane, the bf16x3 kernels (dword loads of the activations, 16-byte loads of the split weights from L2) and the stem raw: uncalibrated"}}
json.dump(out, open(P + TAG + "_pmc_traffic.json", "w"), indent=1)
if len(sys.argv) > 2 and sys.argv[2] == "counters":       # stage 1: the counter files only (commit them, THEN run the bench stage)
    print("profiles/%s_pmc_traffic.json, %s_solver_counters.json, %s_step_instructions.txt rebuilt" % (TAG, TAG, TAG))
    sys.exit(0)


# ---- README
def table(path, n=14):
    rr = list(csv.DictReader(open(path)))
    o = ["| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in rr[:n]:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")[:80].replace("|", "/")
        o.append("| `%s` | %s | %.2f | %.1f | %.1f |" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return "\n".join(o)


line, ser, s3 = (json.load(open(P + TAG + n)) for n in ("_bench_line.json", "_bench_line_serial.json", "_bench_line_pipelined.json"))
tr, pm = json.load(open(P + TAG + "_train_line.json")), out
k = line["kernels"]; cv = k["conv2d_kernel"]; pw = k["pointwise_gemm_kernel(+point_head)"]
srows = list(csv.DictReader(open(P + TAG + "_bench_kernel_stats_serial.csv")))


def avg(sub):
    r = [x for x in srows if sub in x["Name"]]
    return sum(float(x["TotalDurationNs"]) for x in r) / max(1, sum(int(x["Calls"]) for x in r)) / 1e3


gt = open(P + TAG + "_gputest_tail.txt").read().strip().splitlines()[-1]
wl = open(P + TAG + "_winograd_layers.txt").read().strip().splitlines()
cvp = pm["conv2d_resnet34_B32_160x512"]; sp = pm["solve_kernel_F32_R60_N20480"]
RN = TAG.lstrip("r").lstrip("0") or "0"
sc = json.load(open(P + TAG + "_solver_counters.json")) if os.path.exists(P + TAG + "_solver_counters.json") else None
sc_txt = ""
if sc:
    flop = 64.0 * (2 * sc["SQ_INSTS_VALU_FMA_F64"] + sc["SQ_INSTS_VALU_ADD_F64"] + sc["SQ_INSTS_VALU_MUL_F64"])
    sc_txt = ("* `%s_solver_counters.json` -- `tools/prof_solver_counters.sh`: SQ / TCP / TCC counters of `solve_kernel` per launch (separate --pmc passes): "
              "%.2e VALU, %.2e SALU, %.2e LDS, %.2e VMEM wave-instructions; fp64 FMA / MUL / ADD %.2e / %.2e / %.2e = %.2e fp64 flop executed per launch; "
              "wave cycles: %.0f %% waiting (s_waitcnt / barrier), %.0f %% issue stalls, %.0f %% issuing; L2 hit rate %.3f; FETCH %.1f MiB raw, WRITE %.1f MiB.\n"
              % (TAG, sc["SQ_INSTS_VALU"], sc["SQ_INSTS_SALU"], sc["SQ_INSTS_LDS"], sc["SQ_INSTS_VMEM_RD"], sc["SQ_INSTS_VALU_FMA_F64"], sc["SQ_INSTS_VALU_MUL_F64"],
                 sc["SQ_INSTS_VALU_ADD_F64"], flop, 100 * sc["SQ_WAIT_ANY"] / sc["SQ_WAVE_CYCLES"], 100 * sc["SQ_WAIT_INST_ANY"] / sc["SQ_WAVE_CYCLES"],
                 100 * sc["SQ_ACTIVE_INST_ANY"] / sc["SQ_WAVE_CYCLES"], sc["TCC_HIT_sum"] / (sc["TCC_HIT_sum"] + sc["TCC_MISS_sum"]), sc["FETCH_SIZE"] / 1024, sc["WRITE_SIZE"] / 1024))
wino_counters_txt = ("* `%s_winograd_counters.txt` -- `tools/prof_winograd.sh`: SQ / TA / TCP / TCC counters of `wino_conv_kernel` on the stage-3 shape.\n" % TAG
                     if os.path.exists(P + TAG + "_winograd_counters.txt") else "")
extra_txt = ""
if os.path.exists(P + TAG + "_EXTRA.md"):        # the round's A/B calls (tools/oneoff/<tag>_call*.sh), described by hand
    extra_txt = open(P + TAG + "_EXTRA.md").read().rstrip() + "\n"
txt = f"""# Round-{RN} profiles (1x MI355X, ROCm 7.2)

Everything here was produced by ONE gpurun call of `tools/final_round.sh {TAG}` (GPU tests, `bench.py`, `bench.py --mode train`, the
micro-benchmarks, `tools/profile_round.sh {TAG}`) at the end of the round and turned into this directory by `tools/make_profiles.py`;
files listed in KEEP_OLD (if present) come from earlier calls of the round.

* `{TAG}_gputest_tail.txt` -- `python -m pytest tests -q -m gpu`: {gt}.
* `{TAG}_bench_line.json` -- the JSON line of `python bench.py` ({line['steps']} steps, {line['warmup']} warm-up, {line['config']['streams']} streams on {line['config']['hw_queues']} hardware queues, one hipGraph per stream, CPU baseline
  leg included): **{line['value']:.0f} frames/s** resident ({line['ms_per_step']:.2f} ms per 32-frame step), {line['value_with_h2d']:.0f} frames/s with the
  host->device copy of every batch inside the step; device time of one batch {line['latency_ms_per_batch']['streams_%d' % line['config']['streams']]:.1f} ms with all streams busy,
  {line['latency_ms_per_batch']['one_step_in_flight']:.1f} ms with a single step in flight.  Roofline object = time-dominant family = `solve_kernel`:
  {k['solve_kernel']['ms_per_step']:.2f} ms per step, {k['solve_kernel']['achieved']:.1f} TFLOP/s by SURVEY 8(d)'s unit = {k['solve_kernel']['frac']:.2f} of the 78.6 TFLOP/s fp64 vector peak.
  Per family (`kernels`): convolution {cv['ms_per_step']:.2f} ms = {cv['achieved_reference_algorithmic']:.1f} TFLOP/s algorithmic; **{cv['frac']:.2f}** of the matrix pipes it runs on (executed products at peak rate / time; {cv['bf16x3']['calls_per_step']} bf16x3
  launches {cv['bf16x3']['ms_per_step']:.2f} ms = {cv['bf16x3']['fp32_equivalent_tflops']:.0f} TFLOP/s fp32-equivalent = {cv['bf16x3']['executed_bf16_tflops']:.0f} TFLOP/s of executed bf16 products = {cv['bf16x3']['frac_of_bf16_mfma_peak']:.2f} of the 2.5 PFLOP/s bf16 peak;
  {cv['winograd']['calls_per_step']} Winograd launches {cv['winograd']['ms_per_step']:.2f} ms; other implicit-GEMM launches {cv['winograd']['direct_kernel_ms_per_step']:.2f} ms; stem
  {cv['winograd']['stem_kernel_ms_per_step']:.2f} ms); pointwise {pw['ms_per_step']:.2f} ms ({pw['achieved_executed']:.0f} TFLOP/s executed = {pw['frac']:.2f} of the matrix pipes its launches run on; {pw['achieved_reference_algorithmic']:.0f} TFLOP/s by the reference's layer sizes); index_max
  {k['index_max_kernel']['ms_per_step']*1e3:.0f} us in-pipeline ({k['index_max_kernel']['achieved']:.0f} GB/s).  CPU baseline: {line['cpu_baseline']['value']:.3f} frames/s on {line['cpu_baseline']['cores']} threads.
* `{TAG}_bench_kernel_stats_pipelined.csv` / `{TAG}_bench_line_pipelined.json` -- `rocprofv3 --kernel-trace --stats --output-format csv --
  python bench.py --no-cpu-baseline --no-h2d-pass` ({s3['value']:.0f} frames/s under the profiler); several batches in flight: durations include
  contention.
* `{TAG}_bench_kernel_stats_serial.csv` / `{TAG}_bench_line_serial.json` -- the same with `--streams 1` ({ser['value']:.0f} frames/s): durations without
  contention.  `solve_kernel` averages {avg('solve_kernel')/1e3:.2f} ms (rocprof) against {ser['kernels']['solve_kernel']['ms_per_step']:.2f} ms from the HIP events of the same run's JSON
  line; `conv3x3_x3_kernel` (all instances) {avg('conv3x3_x3_kernel'):.1f} us per call; `point_head_x3_kernel` {avg('point_head_x3_kernel'):.1f} us; `knn_nodes_kernel<3>` {avg('knn_nodes_kernel<3>'):.1f} us, `index_max` {avg('index_max'):.1f} us.
* `{TAG}_pmc_conv_{{FETCH,WRITE}}_SIZE.csv`, `{TAG}_pmc_traffic.json` -- separate `--pmc` passes (kernel-trace only) on `tools/bench_conv.py`; the
  solver's entry is COPIED from `{TAG}_solver_counters.json` (one source for its traffic: the file the bench line's `counters_file` names).
  solve_kernel: {sp['FETCH_SIZE_KiB']/1024:.1f} MiB fetched raw (x2 = {sp['fetch_bytes_corrected']/1e6:.1f} MB) per launch against 10.8 MB
  of once-through records + boxes (the working set is cache resident), {sp['WRITE_SIZE_KiB']/1024:.1f} MiB written (the classification-cache entries, 16 B per missed
  cluster and sweep: the L2 is write-through, every store leaves it).  Convolution family of one
  encoder pass: FETCH {cvp['FETCH_SIZE_KiB_per_encoder_pass_raw']/1024:.0f} MiB raw / {cvp['FETCH_SIZE_KiB_per_encoder_pass_corrected']/1024:.0f} MiB corrected, WRITE {cvp['WRITE_SIZE_KiB_per_encoder_pass_raw']/1024:.0f} MiB = {cvp['hbm_bytes_per_call_corrected']/1e6:.0f} MB per convolution call against
  52 MB compulsory.  index_max (fresh passes on cold inputs): C = 64 {pm['index_max_C64_B32_N20480_K128']['hbm_bytes_corrected']/1e6:.0f} MB against {pm['index_max_C64_B32_N20480_K128']['algorithmic_bytes']/1e6:.0f} MB algorithmic, C = 32 {pm['index_max_C32_B32_N20480_K128']['hbm_bytes_corrected']/1e6:.0f} / {pm['index_max_C32_B32_N20480_K128']['algorithmic_bytes']/1e6:.0f} MB.  (`{TAG}_bench_line.json` was produced AFTER
  these counter files were committed: its `traffic` / `frac_executed` fields are computed from them -- `counters_file` in the line names the file.)
* `{TAG}_conv_layers.txt` -- `tools/bench_conv_x3.py`: per 3x3 layer shape at B = 32, the bf16x3 direct convolution (every tile configuration that
  runs the shape) against the fp32-MFMA kernels it replaces (Winograd; for stride 2 the direct kernel + the 1x1 branch), then the whole image
  encoder under the `conv_x3` masks:
```
{chr(10).join(open(P + TAG + "_conv_layers.txt").read().strip().splitlines())}
```
* `{TAG}_convx3_counters.json` -- `tools/prof_kernel_counters.sh` on `tools/run_conv_x3_only.py`: per kernel instance (= layer shape) the matrix-pipe busy
  cycles (`SQ_VALU_MFMA_BUSY_CYCLES`, cycles summed over the SIMDs) against the waves' lifetime (`SQ_WAVE_CYCLES`, quad-cycles x 4), waits, LDS bank
  conflicts, VALU / LDS instructions per matrix instruction, FETCH / WRITE.
* `{TAG}_stem_x3.txt`, `{TAG}_head_x3.txt` -- `tools/bench_stem_x3.py` (conv1 + bn1 + relu + max-pool as one launch on the bf16 matrix instructions against
  the two fp32-MFMA launches) and `tools/bench_head_x3.py` (the coarse per-point head: eight / four waves per workgroup, tables in LDS / from memory;
  alternating rounds -- the first timings of a process run at lower clocks):
```
{open(P + TAG + "_stem_x3.txt").read().strip() if os.path.exists(P + TAG + "_stem_x3.txt") else "(not collected)"}
{open(P + TAG + "_head_x3.txt").read().strip() if os.path.exists(P + TAG + "_head_x3.txt") else "(not collected)"}
```
* `{TAG}_conv_x3_accuracy.txt` -- `tools/diag_conv_x3_accuracy.py`: error against an fp64 convolution, bf16x3 per configuration next to the fp32-MFMA kernels.
* `{TAG}_mfma_rounding.txt` -- `tools/probe_mfma_rounding.hip`: how the bf16 matrix instructions round (products of one instruction are aligned to the
  largest addend and truncated below its last bit; the fp32-input instruction is an exact fma chain) -- why the bf16x3 kernels keep the small products apart.
* `{TAG}_winograd_layers.txt` -- the fp32 kernels alone (direct vs the Winograd variants on the four stride-1 shapes):
```
{chr(10).join(wl[-7:])}
```
{wino_counters_txt}* `{TAG}_solver_phases.txt` -- `PROF=1 python tools/bench_solver.py`: clock64() phase counters, cluster / line-search statistics, sweep-count
  percentiles.
{sc_txt}{extra_txt}* `{TAG}_call_times.txt` -- `tools/call_times.py`: HIP events around every C-ABI call of one serial step, with the contraction shapes.
* `{TAG}_sweep_streams.txt` -- the headline against streams / hardware queues.
* `{TAG}_step_instructions.txt` -- `tools/prof_step_instructions.sh`: VALU / MFMA / SALU / LDS wave-instruction counters of every kernel of a step
  (what the 8-stream step time is bounded by: DESIGN.md section 4).
* `{TAG}_index_max_cold.txt` -- cache-cold index_max.
* `{TAG}_train_line.json`, `{TAG}_train_kernel_stats.csv` -- `python bench.py --mode train` (reference training configuration: batch 8, 20480
  points, 160x512, coarse+fine): **{tr['ms_per_step']:.1f} ms per step = {tr['value']:.0f} frames/s**,
  and `rocprofv3 --kernel-trace --stats` of the same command.

Top kernels of the bench command (default streams, durations include overlap):

{table(P + TAG + '_bench_kernel_stats_pipelined.csv')}

Same, one batch at a time (`--streams 1`):

{table(P + TAG + '_bench_kernel_stats_serial.csv')}

Training step (`--mode train`, batch 8):

{table(P + TAG + '_train_kernel_stats.csv', 12)}
"""
open(P + TAG + "_README.md", "w").write(txt)
print("profiles/%s_* rebuilt: %.0f frames/s, conv frac %.2f, train %.1f ms" % (TAG, line["value"], cv["frac"], tr["ms_per_step"]))
