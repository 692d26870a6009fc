# Builds the HIP module (C ABI of include/datum_ocean_hip.h) for gfx950, the C++ host shim and the CPU oracle.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function

LIB = datum_amd/lib/libdatum_ocean_hip.so
SRC = datum_amd/csrc/ocean_capi.hip
DEPS = datum_amd/csrc/ocean_kernels.hip datum_amd/csrc/ocean_literal.hip datum_amd/csrc/ocean_gen.hip datum_amd/csrc/ocean_farm.hip datum_amd/csrc/ocean_fft_core.h include/datum_ocean_hip.h

HOSTLIB = datum_amd/lib/libdatum_ocean_host.so
HOSTSRC = datum_amd/host/ocean.cpp datum_amd/host/host_capi.cpp
HOSTDEPS = datum_amd/host/ocean.h datum_amd/host/lml.h include/datum_ocean_hip.h
CXX ?= g++
HOSTFLAGS ?= -O2 -std=c++14 -fPIC -ffp-contract=off -fno-fast-math -Wall

# measurement aid of bench.py --standin-peers (a CU-occupying stand-in for the all-gather on one GPU); not the ocean path
STANDIN = datum_amd/lib/libdatum_farm_standin.so

all: $(LIB) $(HOSTLIB) $(STANDIN) oracle examples

$(STANDIN): datum_amd/csrc/farm_standin.hip
	@mkdir -p datum_amd/lib
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -o $@ datum_amd/csrc/farm_standin.hip

# the C++ host shim links only against the C ABI of the HIP module
$(HOSTLIB): $(HOSTSRC) $(HOSTDEPS) $(LIB)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOSTSRC) -Ldatum_amd/lib -ldatum_ocean_hip -Wl,-rpath,'$$ORIGIN'

$(LIB): $(SRC) $(DEPS)
	@mkdir -p datum_amd/lib
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(SRC)

oracle:
	$(MAKE) -C oracle liboracle.so

# datum's example-ocean flow against the host shim, headless
EXAMPLE = examples/ocean_headless
FARMEXAMPLE = examples/ocean_farm
examples: $(EXAMPLE) $(FARMEXAMPLE)

# the tile farm from C++ alone (N processes, RCCL through the C ABI)
$(FARMEXAMPLE): examples/ocean_farm.cpp $(HOSTLIB)
	$(CXX) -O2 -std=c++14 -Wall -o $@ examples/ocean_farm.cpp -Ldatum_amd/lib -ldatum_ocean_host -ldatum_ocean_hip -Wl,-rpath,'$$ORIGIN/../datum_amd/lib'

$(EXAMPLE): examples/ocean_headless.cpp $(HOSTLIB)
	$(CXX) -O2 -std=c++14 -Wall -o $@ examples/ocean_headless.cpp -Ldatum_amd/lib -ldatum_ocean_host -ldatum_ocean_hip -Wl,-rpath,'$$ORIGIN/../datum_amd/lib'

# CPU emulation of the thread-parallel line FFT (tests only)
EMUL = tests/cpu/libfft_core_emul.so
emul: $(EMUL)

$(EMUL): tests/cpu/fft_core_emul.cpp datum_amd/csrc/ocean_fft_core.h
	$(CXX) -O2 -std=c++14 -fPIC -shared -o $@ tests/cpu/fft_core_emul.cpp

# stand-in for the Vulkan side of the external-memory handshake (GPU tests only)
EXTMEM = tests/gpu/libextmem_helper.so
helpers: $(EXTMEM)

$(EXTMEM): tests/gpu/extmem_helper.hip
	$(HIPCC) -O2 -std=c++17 --offload-arch=$(ARCH) -fPIC -shared -o $@ tests/gpu/extmem_helper.hip

resource-usage: $(SRC) $(DEPS)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage -c -o /dev/null $(SRC) 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" 

clean:
	rm -f $(LIB) $(HOSTLIB) $(STANDIN) $(EMUL) $(EXAMPLE) $(FARMEXAMPLE) $(EXTMEM)
	$(MAKE) -C oracle clean

.PHONY: all oracle emul helpers examples clean resource-usage
