# development loop: rebuild only what changed (ANI-2x factor shape only: -DNNPOPS_ONLY_ANI2X_SHAPE) and run a command on the GPU box
#   bash tools/dev.sh 'python tools/ab.py "" "NNPOPS_X=1"'
set -e
cd "$(dirname "$0")/.."
NNPOPS_HIPCC_FLAGS="-DNNPOPS_ONLY_ANI2X_SHAPE" python -m nnpops_amd.build > /dev/null
gpurun --timeout ${TIMEOUT:-600} -- "export NNPOPS_HIPCC_FLAGS=-DNNPOPS_ONLY_ANI2X_SHAPE; $1" 2>&1 | grep -v "amdgpu.ids" | tail -${TAIL:-30}
