for c in 32 64 128 256 512 1024; do
  python tools/bench_conv.py --shape conv,$c,256,1,1,0,4,64 --reps 30 2>/dev/null | tail -2 | head -1
done
for c in 32 64 128 256 512; do
  python tools/bench_conv.py --shape conv,$c,128,1,1,0,8,192 --reps 30 2>/dev/null | tail -2 | head -1
done
