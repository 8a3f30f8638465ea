# Variants of the command line on the files tests/e2e_cli.py left in /tmp/bf_e2e (run that first, in the same gpurun call)
set -u
D=/tmp/bf_e2e
EXE=breakfast_amd/csrc/breakfast-hip
run() { # label, extra flag ("-" for none), env...
  label=$1; flag=$2; shift 2
  [ "$flag" = "-" ] && flag=
  for i in 1 2 3; do
    s=$(date +%s%N)
    env "$@" $EXE detect --anchor-len=20 --anchor-hits=$D/hits.txt --threads=16 --timing $flag unused.bam $D/g > $D/out.sv 2> $D/err.txt
    e=$(date +%s%N)
    echo "$label run $i wall_ms $(( (e - s) / 1000000 )) total $(grep 'TIMING.total' $D/err.txt | cut -f3) parse $(grep 'parse_and_submit' $D/err.txt | cut -f3) submit $(grep 'of_which_bf_submit' $D/err.txt | cut -f3) gpu $(grep 'gpu_localise' $D/err.txt | cut -f3) fw $(grep 'format_and_write' $D/err.txt | cut -f3) sha $(sha256sum $D/out.sv | cut -c1-12)"
  done
}
run default - BREAKFAST_HIP_PARSER_TIMING=1
grep parser: $D/err.txt
run b16M --block-bytes=16000000 BREAKFAST_HIP_PARSER_TIMING=1
grep parser: $D/err.txt
run g2-p14 - BREAKFAST_HIP_GENOME_THREADS=2 BREAKFAST_HIP_PARSER_THREADS=14 BREAKFAST_HIP_PARSER_TIMING=1
grep parser: $D/err.txt
run s6 - BREAKFAST_HIP_STAGE_THREADS=6 BREAKFAST_HIP_PARSER_TIMING=1
grep parser: $D/err.txt
run ascii --ascii-reads BREAKFAST_HIP_PARSER_TIMING=1
grep parser: $D/err.txt
run teardown - BREAKFAST_HIP_TEARDOWN=1
grep TIMING $D/err.txt
s=$(date +%s%N); $EXE hits /dev/null $D/g > /dev/null 2>&1; e=$(date +%s%N); echo "hits-only (fasta load, no gpu) wall_ms $(( (e - s) / 1000000 ))"
s=$(date +%s%N); $EXE anchors /dev/null > /dev/null 2>&1; e=$(date +%s%N); echo "trivial process (library load only) wall_ms $(( (e - s) / 1000000 ))"
sha256sum $D/gpu.sv | cut -c1-12
