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

HOST_OBJS := $(OUT)/chol_ingest.o $(OUT)/chol_symbolic.o $(OUT)/chol_schedule.o $(OUT)/chol_generate.o
HIP_OBJS  := $(OUT)/chol_kernels.o $(OUT)/chol_kernels_f32.o $(OUT)/chol_api.o

all: $(OUT)/libcholamd.so $(BIN)/cholamd_mmat oracle

$(OUT)/%.o: $(CSRC)/%.c $(CSRC)/chol_plan.h include/cholamd.h
	@mkdir -p $(OUT)
	$(CC) $(CFLAGS) -c $< -o $@

$(OUT)/chol_kernels.o: $(CSRC)/chol_kernels.hip $(CSRC)/chol_plan.h $(CSRC)/chol_kernels.h
	@mkdir -p $(OUT)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

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
asan: $(ASAN_OUT)/libcholamd.so oracle
	CHOLAMD_LIB=$(abspath $(ASAN_OUT)/libcholamd.so) LD_PRELOAD=$$($(CC) -print-file-name=libasan.so):$$($(CC) -print-file-name=libubsan.so) \
	ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 python3 -m pytest tests/test_host.py -x -q -p no:cacheprovider
.PHONY: asan
