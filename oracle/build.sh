#!/bin/bash
# Builds the C half of the CPU oracle (test infrastructure only).  The reference is pure Python,
# so there is no oracle/_ref build: it is imported by tests/golden/make_golden.py instead.
set -e
cd "$(dirname "$0")"
gcc -O2 -fPIC -shared -std=c99 -o libggl_oracle.so ggl_oracle.c -lm
