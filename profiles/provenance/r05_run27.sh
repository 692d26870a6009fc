timeout 900 python -m pytest tests -m gpu -x -q -k "export" 2>&1 | tail -2
echo "== new"; python tools/dbg/pack_time.py 2>&1 | grep export
echo "== previous export kernel"; DATUM_OCEAN_HIP_LIB=$(realpath datum_amd/lib/variants/lib_prev.so) python tools/dbg/pack_time.py 2>&1 | grep export
