# Round-5 LM A/Bs on the GPU box (repo root): third wave for the 8- / 12-camera kernels (TDR_LM_NLDS), cost of the literal
# central-difference build (TDR_LM_LITERAL_FD), and the direction diagnostic with both builds.
O=gpurun_out/r5_lm
mkdir -p $O
python tools/ab_knobs.py 8 1000000 "" TDR_LM_NLDS=4 TDR_LM_NLDS=5 TDR_LM_NLDS=6 > $O/nlds8.log 2>&1; cat $O/nlds8.log
python tools/ab_knobs.py 12 1000000 "" TDR_LM_NLDS=6 > $O/nlds12.log 2>&1; cat $O/nlds12.log
for c in 4 8 16; do python tools/ab_knobs.py $c 1000000 "" TDR_LM_LITERAL_FD=1 > $O/lit$c.log 2>&1; grep -v DIFFER $O/lit$c.log; grep -c DIFFER $O/lit$c.log; done
python tools/ab_knobs.py 2 200000 "" TDR_LM_LITERAL_FD=1 > $O/lit2.log 2>&1; grep -v DIFFER $O/lit2.log
for off in 0 1000 2000; do timeout -k 10 600 python tests/tools/lm_direction_diag.py 1.0 $off > $O/direction_$off.json 2> $O/direction_$off.err && echo diag $off ok; done
