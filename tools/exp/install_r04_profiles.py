"""Copies what tools/collect_profiles.sh + tools/exp/collect_r04_extras.sh left under gpurun_out/final/ into the tracked
profiles/r04_* files (the round's judged artefacts).  usage (this container, repo root): python tools/exp/install_r04_profiles.py"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
R = "r04_"


def cp(src, dst, header=None):
    if not os.path.exists(os.path.join(F, src)):
        print("MISSING", src)
        return
    if header:
        with open(os.path.join(P, dst), "w") as f:
            f.write(header.rstrip("\n") + "\n" + open(os.path.join(F, src)).read())
    else:
        shutil.copyfile(os.path.join(F, src), os.path.join(P, dst))
    print("profiles/" + dst)


cp("bench_default.json", R + "bench_default.json")
cp("kernel_stats.csv", R + "hipgraph_bf16_kernel_stats.csv")
cp("kernel_stats_f32.csv", R + "hipgraph_f32_kernel_stats.csv")
cp("pmc_hbm_traffic_bf16.json", R + "pmc_hbm_traffic_bf16.json")
for dt, how in (("bf16", "rocprofv3 --kernel-trace --stats of: python3 bench.py --cpu-steps 0 --no-roofline --steps 100"),
                ("f32", "rocprofv3 --kernel-trace --stats of: python3 bench.py --dtype f32 --cpu-steps 0 --no-roofline --steps 100")):
    src = os.path.join(P, R + "hipgraph_%s_kernel_stats.csv" % dt)
    if not os.path.exists(src):
        continue
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_table.py"), src], capture_output=True,
                         text=True, check=True).stdout
    with open(os.path.join(P, R + "hipgraph_%s_step_summary.txt" % dt), "w") as f:
        f.write("# python tools/kernel_table.py profiles/%shipgraph_%s_kernel_stats.csv  (%s)\n" % (R, dt, how))
        f.write(out)
    print("profiles/%shipgraph_%s_step_summary.txt" % (R, dt))
    cp("step_timeline_%s.txt" % dt, R + "step_timeline_%s.txt" % dt,
       "# python tools/trace_step.py <the same rocprofv3 run's kernel trace> -v : one steady-state replay of the captured "
       "%s step, kernel by kernel (start, gap to the previous end, duration, end; us; q = HSA queue)" % dt)
cp("microbench_strided_build.txt", R + "microbench_strided_build.txt",
   "# tools/exp/collect_r04_extras.sh: the strided layers' builds ALONE in a replayed graph (no neighbours on the chip), and "
   "the products through the packed table")
cp("event_local_conv_build_knockouts.txt", R + "event_local_conv_build_knockouts.txt",
   "# bash tools/exp/knock_ec.sh 256 : EC_KNOCK timing builds of csrc/evconv.hip (1 = no look-back: ids start at 0 in "
   "every event, 2 = no by-output table, 4 = no table / coordinate stores at all, 8 = no tickets / first flags; sums of bits combine) -- results are wrong, "
   "times only")
cp("microbench_conv_bf16_batch2048.txt", R + "microbench_conv_bf16_batch2048.txt",
   "# python tools/microbench_conv.py 30 bf16 2048")
cp("c5_bf16_kernel_table.txt", R + "c5_bf16_kernel_table.txt")
cp("c5_f32_kernel_table.txt", R + "c5_f32_kernel_table.txt")
if os.path.exists(os.path.join(F, "soak_from_files.json")):      # the tool prints progress lines before its JSON object
    rec = [l for l in open(os.path.join(F, "soak_from_files.json")) if l.startswith("{")]
    with open(os.path.join(P, R + "soak_from_files.json"), "w") as f:
        json.dump(json.loads(rec[-1]), f, indent=1)
    print("profiles/" + R + "soak_from_files.json")
cp("pytest_gpu.log", R + "pytest_gpu.log")
# the other configs: six lines of tools/exp/bench_2d_nets.sh (f32 then bf16: GEP, Ioni, C5), the eval loops, C4
lines = [json.loads(l) for l in open(os.path.join(F, "bench_2d_nets.txt")) if l.startswith("{")]
keys = ["gep_c1_%s (tools/bench_gep.py 256 30 150 0 0.2 %s)", "ioni_%s (tools/bench_ioni.py 256 30 %s)",
        "c5_hybrid_%s (tools/bench_gep.py 256 20 1024 3 0.2 %s)"]
other = {"note": "round 4, one MI355X box: the other BASELINE configs and the inference loops (parity cases with a timing, "
                 "not bench lines); tools/exp/collect_r04_extras.sh"}
assert len(lines) == 6, len(lines)
for i, dt in enumerate(("f32", "bf16")):
    for j, k in enumerate(keys):
        rec = lines[3 * i + j]
        assert ("float32" if dt == "f32" else "bfloat16") in rec["config"], (dt, rec["config"])
        other[k % (dt, dt)] = rec
other["eval_loops (tools/bench_eval.py)"] = json.load(open(os.path.join(F, "bench_eval.json")))
other["c4_deep_f16 (bench.py --config config/psd_c4_deep_fp16.json --samples 512 --dtype f16 --cpu-steps 4)"] = \
    json.load(open(os.path.join(F, "c4_f16.json")))
with open(os.path.join(P, R + "other_configs.json"), "w") as f:
    json.dump(other, f, indent=1)
print("profiles/" + R + "other_configs.json")
