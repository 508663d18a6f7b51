# same-session comparison of two builds of the library: base, new, base, new
set -e
ROOT=$(pwd)
for i in 1 2; do
for lib in base hip; do
EEGGAN_LIB=$ROOT/eeg-gan_amd/libeeggan_$lib.so timeout -k 10 200 python bench.py --no-cpu-baseline $AB_ARGS > gpurun_out/ab_$lib.log 2>&1 || { tail -20 gpurun_out/ab_$lib.log; exit 1; }
tail -1 gpurun_out/ab_$lib.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
done
