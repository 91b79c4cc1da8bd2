// ipc_probe.hip -- can two PROCESSES on one GPU share device memory and events through HIP IPC on this box?
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe
// parent: allocates, fills, exports memory + event handles through a pipe; child: opens them, waits on the event in a stream,
// copies D2D from the mapped memory, checks, writes into the parent's buffer, records its own event back.
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e = (x);                                                                            \
        if (e != hipSuccess) {                                                                         \
            printf("[%d] HIP error %s (%d) at %s:%d\n", (int)getpid(), hipGetErrorString(e), (int)e, __FILE__, __LINE__); \
            fflush(stdout);                                                                            \
            _exit(3);                                                                                  \
        }                                                                                              \
    } while (0)

__global__ void k_fill(double *p, int n, double a)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = a + i;
}

struct Msg {
    hipIpcMemHandle_t mem;
    hipIpcEventHandle_t ev;
    int ok_ev;
};

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1 << 20;  // doubles
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    pid_t pid = fork();
    if (pid == 0) {  // child: HIP is initialised after the fork
        close(p2c[1]);
        close(c2p[0]);
        CK(hipSetDevice(0));
        Msg m;
        if (read(p2c[0], &m, sizeof(m)) != (ssize_t)sizeof(m)) _exit(4);
        double *remote = nullptr;
        CK(hipIpcOpenMemHandle((void **)&remote, m.mem, hipIpcMemLazyEnablePeerAccess));
        hipStream_t st;
        CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if (m.ok_ev) {
            hipEvent_t ev;
            hipError_t e = hipIpcOpenEventHandle(&ev, m.ev);
            printf("child: hipIpcOpenEventHandle -> %s\n", hipGetErrorString(e));
            if (e == hipSuccess) CK(hipStreamWaitEvent(st, ev, 0));
        }
        double *mine = nullptr;
        CK(hipMalloc(&mine, sizeof(double) * n));
        CK(hipMemcpyAsync(mine, remote, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
        std::vector<double> h(n);
        CK(hipMemcpyAsync(h.data(), mine, sizeof(double) * n, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += (h[i] != 7.0 + i);
        printf("child: read through the mapped handle, %d wrong of %d\n", bad, n);
        hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, st, remote, n, 100.0);  // write into the parent's memory
        CK(hipStreamSynchronize(st));
        CK(hipIpcCloseMemHandle(remote));
        char c = bad ? 'F' : 'K';
        if (write(c2p[1], &c, 1) != 1) _exit(5);
        fflush(stdout);
        _exit(bad ? 2 : 0);
    }
    close(p2c[0]);
    close(c2p[1]);
    CK(hipSetDevice(0));
    double *d = nullptr;
    CK(hipMalloc(&d, sizeof(double) * n));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipLaunchKernelGGL(k_fill, dim3((n + 255) / 256), dim3(256), 0, st, d, n, 7.0);
    Msg m;
    CK(hipIpcGetMemHandle(&m.mem, d));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess));
    CK(hipEventRecord(ev, st));
    hipError_t e = hipIpcGetEventHandle(&m.ev, ev);
    printf("parent: hipIpcGetEventHandle -> %s\n", hipGetErrorString(e));
    m.ok_ev = (e == hipSuccess);
    if (!m.ok_ev) CK(hipStreamSynchronize(st));
    fflush(stdout);
    if (write(p2c[1], &m, sizeof(m)) != (ssize_t)sizeof(m)) return 1;
    char c = 0;
    if (read(c2p[0], &c, 1) != 1) c = '?';
    int status = 0;
    waitpid(pid, &status, 0);
    std::vector<double> h(n);
    CK(hipMemcpy(h.data(), d, sizeof(double) * n, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += (h[i] != 100.0 + i);
    printf("parent: child said %c (exit %d); the child's writes into my buffer: %d wrong of %d\n", c, WEXITSTATUS(status), bad, n);
    return (c == 'K' && bad == 0) ? 0 : 1;
}
