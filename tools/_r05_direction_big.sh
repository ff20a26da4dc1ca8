# the direction diagnostic on a larger sample: scale 3 x 5 seed offsets
O=gpurun_out/r5_lm
mkdir -p $O
for off in 0 1000 2000 3000 4000; do timeout -k 10 900 python tests/tools/lm_direction_diag.py 3.0 $off > $O/direction_s3_$off.json 2> $O/direction_s3_$off.err && echo diag $off ok; done
