set -e
for w in c2 c5 mesh; do echo "== $w"; tools/abv.sh "plain default" 2 --workload $w --steps 5 --warmup 2; done
