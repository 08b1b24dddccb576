run() { env "$@" python3 bench.py --model DeepSense --no-cpu-baseline --no-roofline --no-secondary --steps 60 --warmup 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
# same-box A/B inside the DeepSense step: workgroup slots of the row-ring convolution (FOCAL_LAB_CONV_SLOTS; default = the CU count)
for i in 1 2 3; do for s in 512 320 256 200; do echo "slots $s  $(run FOCAL_LAB_CONV_SLOTS=$s)"; done; done
