set -e
mkdir -p gpurun_out/r3
tools/prof3.sh c3
tools/prof3.sh c3_linear --flags 32
