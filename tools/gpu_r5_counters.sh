#!/bin/bash
# developer helper (one gpurun call): SQ instruction / activity counters per kernel for the 4K stream, one picture in flight (the kernels alone on the device), in
# one rocprofv3 --pmc pass (--kernel-trace only: no other trace domain beside the counters) -> gpurun_out/$1/sq_counters.json (copied to profiles/round5_sq_counters.json).
# VALU-issue utilisation = SQ_ACTIVE_INST_VALU * 4 / ( 1024 SIMDs * GRBM_GUI_ACTIVE / 8 ): the gfx94x VALUBusy formula (ROCm 7.2 ships no gfx950 derived metrics);
# rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCCs (8.5 x duration x 2.4 GHz for every kernel), hence the / 8.  Cross-check printed beside it: VALU
# instructions x 4 cycles (a wave64 instruction on a 16-lane SIMD) / ( 1024 SIMDs x duration x 2.4 GHz ).
out=gpurun_out/${1:-r5ctr}; cfg=${2:-4k}; mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--config $cfg --steps 8 --warmup 4 --verify 0 --no-cpu-baseline --streams 1 --host-threads 0 --repeats 1"
SET="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/$out/ctr_$cfg -o c -- python $R/bench.py $A > $R/$out/bench_under_pmc_$cfg.json 2> $R/$out/ctr_$cfg.err) || echo "   (counter pass failed)"
python - "$out" "$cfg" "rocprofv3 --kernel-trace --pmc $SET -- python bench.py $A" <<'PY'
import csv, glob, sys, collections, json
out, cfg, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(out + "/ctr_%s/**/*counter_collection.csv" % cfg, recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
dur = collections.defaultdict(list)
for f in glob.glob(out + "/ctr_%s/**/*kernel_trace.csv" % cfg, recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {}
for k, v in sorted(acc.items()):
    n = max(1, len(disp[k])); w = max(1.0, v["SQ_WAVES"]); gui = max(1.0, v["GRBM_GUI_ACTIVE"]); d = sorted(dur.get(k, [0.0]))
    res[k] = {"launches": n, "us_per_launch_under_pmc_median": round(d[len(d) // 2], 1), "waves_per_launch": round(w / n), "valu_insts_per_wave": round(v["SQ_INSTS_VALU"] / w, 1),
              "salu_insts_per_wave": round(v["SQ_INSTS_SALU"] / w, 1), "lds_insts_per_wave": round(v["SQ_INSTS_LDS"] / w, 1),
              "valu_issue_utilisation": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (gui / 8), 4), "valu_issue_by_instruction_count": round(v["SQ_INSTS_VALU"] / n * 4 / 1024 / max(1.0, d[len(d) // 2] * 2400.0), 4), "wait_inst_any_per_busy_cycle": round(v["SQ_WAIT_INST_ANY"] / max(1.0, v["SQ_BUSY_CYCLES"]), 3),
              "gui_active_cycles_per_launch": round(gui / n)}
json.dump({"_note": "one pass of `%s`; sums over the dispatches of a kernel; valu_issue_utilisation = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCCs) (gfx94x VALUBusy formula; the GRBM counter comes summed over the XCCs), valu_issue_by_instruction_count = VALU instructions x 4 cycles / (1024 SIMDs x median duration x 2.4 GHz), "
           "the kernels run one picture at a time (--streams 1): figures of a kernel alone on the device" % cmd, "kernels": res}, open(out + "/sq_counters_%s.json" % cfg, "w"), indent=1)
for k, r in res.items():
    if k.startswith("k_"): print("%-28s %4d x %7.1f us  waves %7d  VALU/w %7.1f SALU/w %6.1f LDS/w %6.1f  valu_util %.3f" % (k[:28], r["launches"], r["us_per_launch_under_pmc_median"], r["waves_per_launch"], r["valu_insts_per_wave"], r["salu_insts_per_wave"], r["lds_insts_per_wave"], r["valu_issue_utilisation"]))
PY
