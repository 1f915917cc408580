#!/bin/bash
# ASAN + UBSAN pass over the native HOST code: the per-window planning stage and the FASTA parser of the product
# (multiprime_amd/csrc/hostplan.cpp + hostplan_wide.cpp, primerstats.cpp, fasta.cpp — pure host C++) and the plain-C oracle, rebuilt with
# -fsanitize=address,undefined and driven by the CPU test-suite (golden fixtures, fuzzed FASTA files, random alignments,
# world-2 gloo run).  The device kernels cannot be sanitized this way; their out-of-bounds guard is the randomised soak.
# usage: tools/sanitize_host.sh [pytest args]     -> profiles/r0N_sanitizers.txt when run by the author
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/oracle/_build/san
mkdir -p $OUT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
g++ -std=c++17 -shared -fPIC -pthread $SAN -o $OUT/libmprime_host_san.so $ROOT/multiprime_amd/csrc/hostplan.cpp $ROOT/multiprime_amd/csrc/hostplan_wide.cpp \
    $ROOT/multiprime_amd/csrc/primerstats.cpp $ROOT/multiprime_amd/csrc/fasta.cpp
gcc -std=c11 -shared -fPIC $SAN -o $OUT/libmprime_oracle_san.so $ROOT/oracle/mprime_oracle.c -lm
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export MP_HOST_LIB=$OUT/libmprime_host_san.so MP_ORACLE_LIB=$OUT/libmprime_oracle_san.so
cd $ROOT
python -m pytest -q -m "not gpu" -p no:cacheprovider tests/test_host_stage.py tests/test_core_golden.py tests/test_oracle_golden.py tests/test_validate.py tests/test_short_rows.py tests/test_batchfilters.py tests/test_seq_store.py "$@"
# ThreadSanitizer over the threaded parts of the host stage (the parser's parallel join and gather, the planner's worker threads)
unset LD_PRELOAD ASAN_OPTIONS UBSAN_OPTIONS MP_ORACLE_LIB
g++ -std=c++17 -shared -fPIC -pthread -fsanitize=thread -g -O1 -o $OUT/libmprime_host_tsan.so $ROOT/multiprime_amd/csrc/hostplan.cpp $ROOT/multiprime_amd/csrc/hostplan_wide.cpp \
    $ROOT/multiprime_amd/csrc/primerstats.cpp $ROOT/multiprime_amd/csrc/fasta.cpp
LD_PRELOAD="$(gcc -print-file-name=libtsan.so)" TSAN_OPTIONS="halt_on_error=1:report_signal_unsafe=0:exitcode=66" MP_HOST_LIB=$OUT/libmprime_host_tsan.so \
    python -m pytest -q -m "not gpu" -p no:cacheprovider tests/test_host_stage.py "$@"
