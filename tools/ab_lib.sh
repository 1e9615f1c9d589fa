#!/bin/bash
# A/B of builds of libmgs.so in alternating fresh processes (same box, same arena size): usage ab_lib.sh N rounds <other.so>...
N=${1:-512}; rounds=${2:-3}; shift 2
export MGS_ARENA_GB=${MGS_ARENA_GB:-100}
for r in $(seq $rounds); do
  python tools/cycle_time.py $N || exit 1
  for other in "$@"; do MGS_LIBMGS=$other python tools/cycle_time.py $N || exit 1; done
done
