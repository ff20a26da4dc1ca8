# Round 5: why the third wave buys the 8-camera LM kernel 2-4 % and no more — SQ counters of the
# default kernel (196 VGPRs, two waves per SIMD) beside the NLDS = 4 build (168 VGPRs, three waves, four cameras'
# directions in LDS), same stream (8 cameras x 1 M systems, 5 % drop-outs).  -> profiles/r05_lm8_nlds.json
# usage (GPU box, repo root): bash tools/_r05_lm8_pmc.sh <commit>
set -x
export TMPDIR=/tmp
C=${1:-worktree}
O=gpurun_out/prof_lm8
mkdir -p $O
for nl in 0 4; do
  K="TDR_LM_NLDS=$nl"   # (ab_knobs.py resets every knob per setting: the value must come as its argument, not from the environment)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$nl -- python3 tools/ab_knobs.py 8 1000000 $K > $O/stats$nl.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/a$nl -- python3 tools/ab_knobs.py 8 1000000 $K > $O/a$nl.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/b$nl -- python3 tools/ab_knobs.py 8 1000000 $K > $O/b$nl.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/c$nl -- python3 tools/ab_knobs.py 8 1000000 $K > $O/c$nl.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d $O/d$nl -- python3 tools/ab_knobs.py 8 1000000 $K > $O/d$nl.log 2>&1 || echo "pass d failed (counter names)"
  TDR_PROFILE_COMMIT=$C python tools/summarize_sq.py $O/summary$nl.json "ray_kernel<8" 1000000 $O/stats$nl $O/a$nl $O/b$nl $O/c$nl $( [ -d $O/d$nl ] && echo $O/d$nl ) -- "8 cameras x 1 M systems, TDR_LM_NLDS=$nl" | tail -30
done
python3 - <<'PY'
import json
out = {"what": "8-camera LM kernel, 1 M systems with 5 % drop-outs: default build (directions in registers, 196 VGPRs, two waves "
               "per SIMD) against the TDR_LM_NLDS=4 build (last four cameras' directions in LDS, 168 VGPRs, three waves); "
               "rocprofv3 --kernel-trace --stats and separate --pmc passes (tools/_r05_lm8_pmc.sh)"}
for nl in (0, 4):
    out[f"nlds{nl}"] = json.load(open(f"gpurun_out/prof_lm8/summary{nl}.json"))
json.dump(out, open("profiles/r05_lm8_nlds.json", "w"), indent=1)
PY
mkdir -p gpurun_out/profiles_r05 && cp profiles/r05_lm8_nlds.json gpurun_out/profiles_r05/
