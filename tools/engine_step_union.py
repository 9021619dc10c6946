"""The regime the headline runs in, from a rocprofv3 kernel trace of bench.py itself (two actor slices = two encoder launches
in flight on two HIP streams): per env step the UNION of the encoder kernels' [start, end] intervals (the time during which at
least one encoder kernel runs), the sum of their durations, and the split of that sum into the bandwidth-bound front of the
network and its MFMA-bound back (what perfect co-scheduling of one slice's front with the other's back could reach).

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python bench.py --steps 1 --warmup 1 --no-... > line.json
  python tools/engine_step_union.py OUT/.../t_kernel_trace.csv line.json profiles/engine_step_union.json > profiles/r06_engine_step_union.txt

Window: the LAST rollout of the trace (the timed iteration), its first and last `edge` env steps dropped (ramp-up behind the
previous update / drain into the next one).  An env step = one stem launch per slice stream.
"""
import csv
import json
import sys

ENC = ("stem_conv1", "stem7_pool", "conv3x3_rows", "conv3x3_narrow", "conv_igemm_kernel", "conv_igemm8_kernel", "conv1x1_pair", "conv1x1_regw",
       "bneck23_kernel", "conv3x3_img_kernel", "avgpool2_kernel")
# bandwidth-bound families (stem, layer-1 / layer-2 block boundaries, narrow 3x3, pools) vs the MFMA-bound rest
BW = ("stem_conv1", "stem7_pool", "conv3x3_rows", "conv3x3_narrow", "conv1x1_pair", "conv1x1_regw", "avgpool2_kernel")


def is_encoder(name: str) -> bool:
    if not any(k in name for k in ENC):
        return False
    # the policy's compressor conv over stored features runs on the same 8-wave kernel (X3 instances: 5th template argument true)
    if "conv_igemm8_kernel<" in name:
        args = name.split("conv_igemm8_kernel<", 1)[1].split(">", 1)[0].split(",")
        if len(args) >= 5 and args[4].strip() == "true":
            return False
    return True


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return tot + (ce - cs)


def main():
    trace, line_path, out_json = sys.argv[1], sys.argv[2], sys.argv[3]
    edge = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    line = json.loads([l for l in open(line_path) if l.startswith("{")][-1])
    cfg = line["config"]
    T, frames = cfg["rollout"], cfg["actors_per_gpu"]
    n_conc = line["roofline"]["concurrent_launches"]
    rows = [r for r in csv.DictReader(open(trace)) if is_encoder(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    stems = [int(r["Start_Timestamp"]) for r in rows if "stem_conv1" in r["Kernel_Name"] or "stem7_pool" in r["Kernel_Name"]]
    per_it = T * n_conc
    assert len(stems) >= per_it, f"{len(stems)} stem launches in the trace, one rollout has {per_it}"
    last = stems[-per_it:]                                  # the last rollout's stem launches (n_conc per env step)
    t0, t1 = last[edge * n_conc], last[-edge * n_conc]
    steps = (per_it - 2 * edge * n_conc) // n_conc
    win = [(max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1), r["Kernel_Name"]) for r in rows
           if int(r["End_Timestamp"]) > t0 and int(r["Start_Timestamp"]) < t1]
    u = union([(s, e) for s, e, _ in win]) / 1e6 / steps
    tot = sum(e - s for s, e, _ in win) / 1e6 / steps
    bw = sum(e - s for s, e, n in win if any(k in n for k in BW)) / 1e6 / steps
    mf = tot - bw
    period = (t1 - t0) / 1e6 / steps
    flop = 2.0 * 5_367_226_368 * frames
    rec = {"plan_hash": line["roofline"]["plan_hash"], "frames_per_env_step": frames, "concurrent_launches": n_conc, "env_steps": steps,
           "union_ms_per_env_step": round(u, 4), "sum_kernel_ms_per_env_step": round(tot, 4), "overlap_factor": round(tot / u, 3),
           "env_step_period_ms": round(period, 4),
           "sum_bandwidth_bound_ms": round(bw, 4), "sum_mfma_bound_ms": round(mf, 4),
           "co_scheduling_ceiling_ms": round(max(bw, mf), 4),
           "tflops_over_union": round(flop / (u * 1e-3) / 1e12, 1), "frac_of_2500": round(flop / (u * 1e-3) / 1e12 / 2500.0, 4),
           "bench_value_under_profiler": line["value"], "bench_union_ms_live_under_profiler": line["roofline"]["avg_step_union_ms"],
           "source": "rocprofv3 --kernel-trace of bench.py (tools/engine_step_union.py)"}
    json.dump(rec, open(out_json, "w"), indent=1)
    print(f"# engine steady state under rocprofv3 --kernel-trace: {steps} env steps of the last rollout, {frames} frames per env step in {n_conc} "
          f"concurrent launches; plan {rec['plan_hash']}")
    print(f"env step period            {period:8.4f} ms")
    print(f"UNION of encoder kernels   {u:8.4f} ms per env step  -> {rec['tflops_over_union']} TFLOP/s = {rec['frac_of_2500']} of 2.5 PFLOP/s")
    print(f"sum of encoder kernel time {tot:8.4f} ms per env step  (overlap factor {rec['overlap_factor']}: kernels of the two launches run side by side)")
    print(f"  bandwidth-bound families {bw:8.4f} ms  (stem, narrow 3x3, layer-1/2 block boundaries, pools)")
    print(f"  MFMA-bound families      {mf:8.4f} ms  (conv_igemm / conv_igemm8 / conv_bneck / conv3x3_img)")
    print(f"co-scheduling ceiling      {max(bw, mf):8.4f} ms  = max of the two sums: one slice's front running entirely under the other's back")
    print(f"bench.py under the profiler: {line['value']} env-frames/s, live HIP-event union {line['roofline']['avg_step_union_ms']} ms")
    # per kernel family, per env step
    fam = {}
    for s, e, n in win:
        k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        d = fam.setdefault(k, [0, 0])
        d[0] += e - s
        d[1] += 1
    print("# per kernel instance: ms per env step, launches per env step")
    for k, (d, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"  {d / 1e6 / steps:8.4f}  {c / steps:6.1f}  {k}")


if __name__ == "__main__":
    main()
