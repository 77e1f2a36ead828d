set -e
mkdir -p gpurun_out/r4
tools/prof4.sh c3
tools/prof4.sh c3_linear --flags 32
tools/prof4.sh c5 --workload c5
tools/prof4.sh mesh --workload mesh
