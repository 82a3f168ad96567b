# Top-level build: HIP library (gfx950), drop-in executables, oracle (test infrastructure).
HIPCC   ?= /opt/rocm/bin/hipcc
CXX     ?= g++
ARCH    ?= gfx950
ROCM    ?= /opt/rocm
EXTRA   ?=
HIPFLAGS = $(EXTRA) --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Iinclude
LIB      = ebwt2snp_amd/libe2s_hip.so
CSRC     = ebwt2snp_amd/csrc
HOST     = ebwt2snp_amd/host

all: lib bins oracle

lib: $(LIB)

LIBSRC   = $(CSRC)/e2s_kernels.hip $(CSRC)/e2s_tail.hip $(CSRC)/e2s_capi.hip $(HOST)/snp_tail.cpp $(HOST)/fasta_index.cpp $(HOST)/reads_upload.cpp $(HOST)/chunked.cpp
# the library exports what include/e2s_hip.h declares and nothing else (the calls between its own translation units, the kernels' host
# stubs and launchers stay local; e2s_diag_*: diagnostic builds only, e.g. EXTRA=-DE2S_DIAG_STAMPS)
EXPORTS  = build/e2s_exports.map
$(EXPORTS): include/e2s_hip.h
	@mkdir -p build
	@(echo "{ global:"; grep -oE '\be2s_[a-z0-9_]+[[:space:]]*\(' $< | tr -d ' (' | sort -u | sed 's/$$/;/'; echo "e2s_diag_*; local: *; };") > $@
$(LIB): $(LIBSRC) $(CSRC)/e2s_internal.h $(CSRC)/e2s_swar.h $(CSRC)/e2s_plan.h $(CSRC)/e2s_env.h $(HOST)/fasta_index.hpp include/e2s_hip.h $(EXPORTS)
	$(HIPCC) $(HIPFLAGS) -shared -Wl,--version-script=$(EXPORTS) -o $@ $(LIBSRC) -lpthread

bins: bin/ebwt2clust bin/clust2snp bin/ebwt2snp_mgpu

# the multi-GPU driver links RCCL directly (ncclCommInitAll / ncclAllGather) and the HIP runtime for its exchange buffers
bin/ebwt2snp_mgpu: $(HOST)/ebwt2snp_mgpu_main.cpp $(HOST)/egsa_files.hpp $(LIB)
	@mkdir -p bin
	$(CXX) -O2 -std=c++17 -Wall -D__HIP_PLATFORM_AMD__ -Iinclude -I$(HOST) -I$(ROCM)/include -o $@ $< -Lebwt2snp_amd -le2s_hip -L$(ROCM)/lib -lrccl -lamdhip64 -lpthread \
	    -Wl,-rpath,'$$ORIGIN/../ebwt2snp_amd' -Wl,-rpath,$(ROCM)/lib

bin/%: $(HOST)/%_main.cpp $(HOST)/egsa_files.hpp $(LIB)
	@mkdir -p bin
	$(CXX) -O2 -std=c++17 -Wall -Iinclude -o $@ $< -Lebwt2snp_amd -le2s_hip -Wl,-rpath,'$$ORIGIN/../ebwt2snp_amd'

oracle:
	$(MAKE) -C oracle

asm: $(CSRC)/e2s_kernels.hip
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build_kernels.s $(CSRC)/e2s_kernels.hip -Rpass-analysis=kernel-resource-usage

# Sanitizer builds of the host code that walks pointers, maps files and runs worker threads (host tail, FASTA index) and of the oracle's C
# restatement; tests/test_host_logic.py::test_sanitizer_builds_over_goldens runs the goldens through both (E2S_TAIL_THREADS=8).
SAN_SRC = tests/san_driver.cpp $(HOST)/snp_tail.cpp $(HOST)/fasta_index.cpp
sanitize: build/san_asan build/san_tsan
build/san_oracle_asan.o: oracle/e2s_oracle.c oracle/e2s_oracle.h
	@mkdir -p build
	gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -c -o $@ oracle/e2s_oracle.c
build/san_oracle_tsan.o: oracle/e2s_oracle.c oracle/e2s_oracle.h
	@mkdir -p build
	gcc -O1 -g -fsanitize=thread -c -o $@ oracle/e2s_oracle.c
build/san_asan: $(SAN_SRC) build/san_oracle_asan.o include/e2s_hip.h
	$(CXX) -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -o $@ $(SAN_SRC) build/san_oracle_asan.o -lpthread
build/san_tsan: $(SAN_SRC) build/san_oracle_tsan.o include/e2s_hip.h
	$(CXX) -O1 -g -std=c++17 -fsanitize=thread -Iinclude -o $@ $(SAN_SRC) build/san_oracle_tsan.o -lpthread

clean:
	rm -rf $(LIB) bin build_kernels.s build/san_*
	$(MAKE) -C oracle clean

.PHONY: all lib bins oracle asm clean sanitize
