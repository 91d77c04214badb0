#!/bin/bash
# A/B of two builds of libmi355nrphy.so on ONE box: profiles/ab_variants.sh with two variants.
# Usage (GPU box, repository root): bash profiles/ab_lib.sh build/variants/old.so build/variants/new.so [rounds]
exec bash "$(dirname "$0")/ab_variants.sh" "${3:-3}" "$1" "$2"
