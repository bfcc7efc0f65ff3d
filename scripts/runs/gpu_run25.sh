set -o pipefail
mkdir -p gpurun_out
( while true; do sleep 45; date >> gpurun_out/r2x.alive; done ) & alive=$!
trap "kill $alive 2>/dev/null" EXIT
echo "== host path alone"; python3 scripts/trace_host.py | tail -3 || exit 1
echo "== host path after a resident map (kept)"; python3 scripts/trace_host.py after_resident | tail -3 || exit 1
echo "== host path after a resident map (closed)"; python3 scripts/trace_host.py after_resident close | tail -3 || exit 1
timeout -k 10 300 python3 scripts/c3_two_pass.py || exit 1
