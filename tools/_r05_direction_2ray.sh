# the two-ray population alone, 16 x 2.4 M systems (other seeds than every earlier sample): is the GPU-vs-oracle residue real?
O=gpurun_out/r5_lm
mkdir -p $O
for off in 5000 6000 7000 8000 9000 10000 11000 12000 13000 14000 15000 16000 17000 18000 19000 20000; do
  timeout -k 10 600 python tests/tools/lm_direction_diag.py 10.0 $off 2 > $O/direction_2ray_$off.json 2> $O/direction_2ray_$off.err && echo diag $off ok $(date +%s)
done
