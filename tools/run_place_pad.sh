# Small concurrent launches and where their workgroups land (launch.hpp: place_pad): PGPU_PLACE_PAD=0 is the round-5 behaviour,
# the default (192) pads launches of at most 192 workgroups of the multi-lane forms to one workgroup per CU.
B=pailliercryptolib_amd/ipcl_api_bench
O=gpurun_out/r06_place_pad.txt
: > $O
for rep in 1 2; do
for pad in 0 192; do
  for n in 256 700 1024 2048; do for t in 1 2 4; do echo "PGPU_PLACE_PAD=$pad" >> $O; PGPU_PLACE_PAD=$pad timeout 120 $B --threads $t $n 150 >> $O 2>&1; done; done
  for n in 512 1024 2048; do for t in 1 2 4; do echo "PGPU_PLACE_PAD=$pad" >> $O; PGPU_PLACE_PAD=$pad timeout 120 $B --threads-mul $t $n 80 >> $O 2>&1; done; done
done
done
