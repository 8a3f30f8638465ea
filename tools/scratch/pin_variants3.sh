set -u
D=/tmp/bf_e2e
EXE=breakfast_amd/csrc/breakfast-hip
run() { # label, extra flag ("-" for none), env...
  label=$1; flag=$2; shift 2
  [ "$flag" = "-" ] && flag=
  for i in 1 2 3; do
    sleep 1
    : > $D/out.sv
    s=$(date +%s%N)
    env "$@" $EXE detect --anchor-len=20 --anchor-hits=$D/hits.txt --threads=16 --timing $flag unused.bam $D/g > $D/out.sv 2> $D/err.txt
    e=$(date +%s%N)
    echo "$label run $i wall_ms $(( (e - s) / 1000000 )) total $(grep 'TIMING.total' $D/err.txt | cut -f3) parse $(grep 'parse_and_submit' $D/err.txt | cut -f3) submit $(grep 'of_which_bf_submit' $D/err.txt | cut -f3) gtail $(grep 'genome_to_hbm_tail' $D/err.txt | cut -f3) gstream $(grep 'genome_stream' $D/err.txt | cut -f3) sha $(sha256sum $D/out.sv | cut -c1-12)"
  done
}
run default - A=1
run g2 - BREAKFAST_HIP_GENOME_THREADS=2
run g2p12 - BREAKFAST_HIP_GENOME_THREADS=2 BREAKFAST_HIP_PARSER_THREADS=12
run g4p10 - BREAKFAST_HIP_GENOME_THREADS=4 BREAKFAST_HIP_PARSER_THREADS=10
run g3p12 - BREAKFAST_HIP_GENOME_THREADS=3 BREAKFAST_HIP_PARSER_THREADS=12
