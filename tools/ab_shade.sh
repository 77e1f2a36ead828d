set -e
tools/abv.sh "default old noshp nopp" 2 --workload c5
tools/abv.sh "default old" 2 --workload c3
tools/abv.sh "default old" 2 --workload mesh
tools/abv.sh "default old" 1 --workload c2
