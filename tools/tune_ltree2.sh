set -e
tools/abv.sh "default s6 s10" 2   # variants: RT_LIB_VARIANT=s6 RT_EXTRA_HIPCC_FLAGS="-DRT_STEPS_PER_CHECK_LTREE=6 -DRT_MAXL_LTREE=10" python -m ray_tracer_s8_amd.build (s10 likewise)
for r in 1 3; do echo "refill $r"; RT_REFILL_EIGHTHS=$r tools/abv.sh "default s6 s10" 1; done
