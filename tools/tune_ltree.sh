set -e
tools/abv.sh "default s6 s10 s11" 1
for r in 1 3 4; do echo "refill $r"; RT_REFILL_EIGHTHS=$r tools/abv.sh "default s10" 1; done
