set -e
tools/abv.sh "default s6 s10" 2
for r in 1 3; do echo "refill $r"; RT_REFILL_EIGHTHS=$r tools/abv.sh "default s6 s10" 1; done
