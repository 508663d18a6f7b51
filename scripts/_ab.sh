# same-session comparison of builds of the library (AB_LIBS: names x of eeg-gan_amd/libeeggan_x.so), two rounds
set -e
ROOT=$(pwd)
for i in 1 2; do
for lib in ${AB_LIBS:-base hip}; do
EEGGAN_LIB=$ROOT/eeg-gan_amd/libeeggan_$lib.so timeout -k 10 200 python bench.py --no-cpu-baseline $AB_ARGS > gpurun_out/ab_$lib.log 2>&1 || { tail -20 gpurun_out/ab_$lib.log; exit 1; }
tail -1 gpurun_out/ab_$lib.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
done
