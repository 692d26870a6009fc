timeout 900 python -m pytest tests -m gpu -x -q -k "pack or farm or gather or payload or example or collective or consumer" 2>&1 | tail -2
python tools/dbg/pack_time.py 2>&1 | grep -v amdgpu
