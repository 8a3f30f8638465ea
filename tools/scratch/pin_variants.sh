# pinned parser batches against pageable ones, and block sizes (after tests/e2e_cli.py, same gpurun call)
set -u
D=/tmp/bf_e2e
EXE=breakfast_amd/csrc/breakfast-hip
run() { # label, extra flag ("-" for none), env...
  label=$1; flag=$2; shift 2
  [ "$flag" = "-" ] && flag=
  for i in 1 2 3; do
    sleep 1
    s=$(date +%s%N)
    env "$@" $EXE detect --anchor-len=20 --anchor-hits=$D/hits.txt --threads=16 --timing $flag unused.bam $D/g > $D/out.sv 2> $D/err.txt
    e=$(date +%s%N)
    echo "$label run $i wall_ms $(( (e - s) / 1000000 )) total $(grep 'TIMING.total' $D/err.txt | cut -f3) parse $(grep 'parse_and_submit' $D/err.txt | cut -f3) submit $(grep 'of_which_bf_submit' $D/err.txt | cut -f3) $(grep parser: $D/err.txt) sha $(sha256sum $D/out.sv | cut -c1-12)"
  done
}
run pageable - BREAKFAST_HIP_PAGEABLE_BATCHES=1 BREAKFAST_HIP_PARSER_TIMING=1
run pinned32M - BREAKFAST_HIP_PARSER_TIMING=1
run pinned16M --block-bytes=16000000 BREAKFAST_HIP_PARSER_TIMING=1
run pinned8M --block-bytes=8000000 BREAKFAST_HIP_PARSER_TIMING=1
run pinned4M --block-bytes=4000000 BREAKFAST_HIP_PARSER_TIMING=1
run pinned2M --block-bytes=2000000 BREAKFAST_HIP_PARSER_TIMING=1
run pageable8M --block-bytes=8000000 BREAKFAST_HIP_PAGEABLE_BATCHES=1 BREAKFAST_HIP_PARSER_TIMING=1
