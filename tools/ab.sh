# A/B of libepi variants on one workload: tools/ab.sh <workload> <variant> ... ("base" = the in-tree library, others build/libepi_<variant>.so)
w=$1; shift
for v in "$@"; do
  if [ "$v" = base ]; then unset EPI_LIB; else export EPI_LIB=$PWD/build/libepi_$v.so; fi
  for rep in 1 2; do
    timeout -k 10 300 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$v.$rep.json 2> gpurun_out/ab_$v.$rep.err || exit 1
  done
done
