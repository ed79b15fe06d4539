python tools/flood_prof.py 2>&1 | grep -v amdgpu.ids | tail -6
python -m pytest tests -m gpu -x -q -k "flood or dense or Dense" 2>&1 | grep -E "passed|failed|error" | tail -3
