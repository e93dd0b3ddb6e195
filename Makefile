# Builds libcholamd.so (C host code + HIP kernels for gfx950), the mmat-compatible CLI and the
# CPU oracle (test infrastructure).  No cmake needed: gcc for the C host code, hipcc for HIP.
HIPCC   ?= /opt/rocm/bin/hipcc
CC      ?= gcc
ARCH    ?= gfx950
CSRC    := cholesky_amd/csrc
OUT     := cholesky_amd/lib
BIN     := cholesky_amd/bin
CFLAGS  := -O2 -fPIC -Wall -Wextra -std=gnu11 -Iinclude -I$(CSRC)
HIPFLAGS:= -O3 -fPIC --offload-arch=$(ARCH) -Iinclude -I$(CSRC) -Wall
# the TRSM strips' step loop (trsm_rr_body, up to 20 column tiles) must unroll completely: its register tiles are indexed by the step; beyond
# LLVM's default pragma-unroll budget the loop stays rolled and the tiles go to scratch (160 B per lane)
KERNFLAGS := -mllvm -pragma-unroll-threshold=100000

HOST_OBJS := $(OUT)/chol_ingest.o $(OUT)/chol_symbolic.o $(OUT)/chol_schedule.o $(OUT)/chol_generate.o
HIP_OBJS  := $(OUT)/chol_kernels.o $(OUT)/chol_kernels_f32.o $(OUT)/chol_api.o

all: $(OUT)/libcholamd.so $(BIN)/cholamd_mmat oracle

$(OUT)/%.o: $(CSRC)/%.c $(CSRC)/chol_plan.h include/cholamd.h
	@mkdir -p $(OUT)
	$(CC) $(CFLAGS) -c $< -o $@

$(OUT)/chol_kernels.o: $(CSRC)/chol_kernels.hip $(CSRC)/chol_plan.h $(CSRC)/chol_kernels.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) $(KERNFLAGS) -c $< -o $@

$(OUT)/chol_kernels_f32.o: $(CSRC)/chol_kernels_f32.hip $(CSRC)/chol_plan.h $(CSRC)/chol_kernels.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(OUT)/chol_api.o: $(CSRC)/chol_api.cpp $(CSRC)/chol_plan.h $(CSRC)/chol_kernels.h include/cholamd.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(OUT)/libcholamd.so: $(HOST_OBJS) $(HIP_OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) -o $@ $^ -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib

$(BIN)/cholamd_mmat: $(CSRC)/mmat_main.c $(OUT)/libcholamd.so include/cholamd.h
	@mkdir -p $(BIN)
	$(CC) $(CFLAGS) -o $@ $< -L$(OUT) -lcholamd -Wl,-rpath,'$$ORIGIN/../lib' -lm

oracle:
	$(MAKE) -s -C oracle

clean:
	rm -rf $(OUT) $(BIN)
	$(MAKE) -s -C oracle clean

.PHONY: all oracle clean

# ---- sanitizer build of the HOST code (ingest, symbolic phase, schedules, generator, C-ABI glue): AddressSanitizer +
# UBSan; the device code is built as usual (GPU sanitizers are not available on the pool).  `make asan` builds
# cholesky_amd/lib/asan/libcholamd.so and runs the CPU test-suite of the host logic against it.
ASAN_OUT := cholesky_amd/lib/asan
SAN := -fsanitize=address,undefined -fno-omit-frame-pointer -g
ASAN_HOST_OBJS := $(ASAN_OUT)/chol_ingest.o $(ASAN_OUT)/chol_symbolic.o $(ASAN_OUT)/chol_schedule.o $(ASAN_OUT)/chol_generate.o
$(ASAN_OUT)/%.o: $(CSRC)/%.c $(CSRC)/chol_plan.h include/cholamd.h
	@mkdir -p $(ASAN_OUT)
	$(CC) $(CFLAGS) -O1 $(SAN) -c $< -o $@
$(ASAN_OUT)/chol_api.o: $(CSRC)/chol_api.cpp $(CSRC)/chol_plan.h $(CSRC)/chol_kernels.h include/cholamd.h
	@mkdir -p $(ASAN_OUT)
	$(HIPCC) $(HIPFLAGS) -O1 $(SAN) -fno-sanitize=function -fno-gpu-sanitize -x hip -c $< -o $@
$(ASAN_OUT)/libcholamd.so: $(ASAN_HOST_OBJS) $(ASAN_OUT)/chol_api.o $(OUT)/chol_kernels.o $(OUT)/chol_kernels_f32.o
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(SAN) -fno-gpu-sanitize -o $@ $^ -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
# leak detection: ON, in a process of its own without an interpreter (tests/native/host_leak.c: plans, schedules of every level and
# world size, the program builder and its self-check, the generator, error paths); the only suppression is the HIP runtime's own
# process-lifetime state (scripts/lsan.supp).  Under pytest the interpreter's and torch's allocations would drown (or, suppressed by
# frame, hide) the library's, so that run keeps detect_leaks=0.
G := tests/golden
$(ASAN_OUT)/host_leak: tests/native/host_leak.c $(ASAN_OUT)/libcholamd.so
	$(HIPCC) -x c -O1 -Iinclude $(SAN) -fno-sanitize=function -o $@ $< -L$(ASAN_OUT) -lcholamd -Wl,-rpath,$(abspath $(ASAN_OUT)) -lm
asan: $(ASAN_OUT)/libcholamd.so $(ASAN_OUT)/host_leak oracle
	ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 LSAN_OPTIONS=suppressions=$(abspath scripts/lsan.supp):print_suppressions=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
	$(ASAN_OUT)/host_leak $(G)/lapl_9x9/lapl_3_2.mtx $(G)/lapl_9x9/lapl_3_2_ord_2.txt $(G)/lapl_9x9/lapl_3_2_clust_2.txt \
	  $(G)/lapl_400x400/lapl_20_2.mtx $(G)/lapl_400x400/lapl_20_2_ord_5.txt $(G)/lapl_400x400/lapl_20_2_clust_5.txt \
	  $(G)/lapl_3375x3375/lapl_15_3.mtx $(G)/lapl_3375x3375/lapl_15_3_ord_5.txt $(G)/lapl_3375x3375/lapl_15_3_clust_5.txt
	CHOLAMD_LIB=$(abspath $(ASAN_OUT)/libcholamd.so) LD_PRELOAD=$$($(CC) -print-file-name=libasan.so):$$($(CC) -print-file-name=libubsan.so) \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python3 -m pytest tests/test_host.py -x -q -p no:cacheprovider
.PHONY: asan
