import sys; sys.path.insert(0,'.')
from datum_amd import capi
for N in (2048,4096):
    try:
        oc=capi.Ocean(N,1); print(N,'ok'); oc.close()
    except Exception as e: print(N,e)
