run() { env $1 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
for s in "A=1" "RCGAN_WGRAD_GROUP_PXMAX=4096" "RCGAN_WGRAD_GROUP_PXMAX=8192" "RCGAN_P8N_MINBLK=128" "RCGAN_P8N_MINBLK=256" "RCGAN_KS2_MAXBLK=512" "RCGAN_KS2_MAXBLK=1024" "RCGAN_KS4_MAXBLK=256" "RCGAN_KS4_MAXBLK=512" "RCGAN_WGRAD_IMG_WGS=64" "RCGAN_WGRAD_IMG_WGS=128" "RCGAN_T128_MINBLK=256" "RCGAN_T128_MINBLK=512" "RCGAN_RIDE_MAXWG=128" "RCGAN_RIDE_MAXWG=512" "RCGAN_P8_MINBLK=128"; do
  echo "$s  $(run "$s")"
done; done
