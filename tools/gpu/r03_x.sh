#!/bin/bash
# round 3, call X: what HIP start-up costs a process that links nothing of ours, against the CLI's first GPU call
set -e
export TMPDIR=/tmp
cat > /tmp/hipmin.cpp <<'CPP'
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int *p) { p[0] = 1; }
int main()
{
	auto t0 = std::chrono::steady_clock::now();
	auto ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	hipSetDevice(0); printf("hipSetDevice %.1f ms\n", ms());
	hipStream_t s; hipStreamCreate(&s); printf("stream %.1f ms\n", ms());
	int *d; hipMalloc((void **)&d, 64); printf("malloc %.1f ms\n", ms());
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s); printf("first kernel %.1f ms\n", ms());
	return 0;
}
CPP
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value /tmp/hipmin.cpp -o /tmp/hipmin 2>/dev/null
for i in 1 2 3; do a=$(date +%s%N); /tmp/hipmin; b=$(date +%s%N); echo "process $(( (b - a) / 1000000 )) ms"; done
cd $GRAFT_REPO_ROOT
printf '>a\nLEAGTLDK\n>b\nMEAGTQDK\n' > /tmp/t.fa
python3 tools/cli_latency.py 2>&1 | cut -c1-120
for i in 1 2; do AT_CLI_TRACE=1 aligntools/c_amd/bin/alignTools local -m 2 -u -2 -o -5 -e -2 /tmp/t.fa 2>&1 | grep -i "trace" | cut -c1-100; done
