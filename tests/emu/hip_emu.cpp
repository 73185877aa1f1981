// Fiber scheduler behind tests/emu/hip_emu.h (test infrastructure only).
#include "hip_emu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
namespace {
struct Fiber {
    ucontext_t ctx;
    bool done;
};
constexpr size_t kStack = 256 * 1024;
constexpr int kMaxThreads = 1024;
ucontext_t g_sched;
std::vector<Fiber> g_fibers;
char* g_stacks = nullptr;
const std::function<void()>* g_body = nullptr;
int g_cur = 0, g_nthreads = 0, g_alive = 0;
long g_progress = 0;
int g_bar_count = 0;
long g_bar_gen = 0;
int g_wave_count[kMaxThreads / 64];
long g_wave_gen[kMaxThreads / 64];
alignas(16) char g_wave_buf[kMaxThreads / 64][64 * 16];
std::vector<char> g_smem;

void set_tid(int i) {
    threadIdx.x = i % blockDim.x;
    threadIdx.y = (i / blockDim.x) % blockDim.y;
    threadIdx.z = i / (blockDim.x * blockDim.y);
}
void yield() {
    int me = g_cur;
    swapcontext(&g_fibers[me].ctx, &g_sched);
}
void trampoline() {
    (*g_body)();
    g_fibers[g_cur].done = true;
    ++g_progress;
    // as on the hardware, a finished wave no longer takes part in s_barrier
    --g_alive;
    if (g_bar_count > 0 && g_bar_count >= g_alive) {
        g_bar_count = 0;
        ++g_bar_gen;
    }
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}
}  // namespace

int lane() { return g_cur & 63; }
int wave_lanes() {
    int w = g_cur >> 6;
    return std::min(64, g_nthreads - w * 64);
}
void* wave_buf() { return g_wave_buf[g_cur >> 6]; }
void* dyn_smem() { return g_smem.data(); }

void block_barrier() {
    long gen = g_bar_gen;
    ++g_progress;
    if (++g_bar_count >= g_alive) {
        g_bar_count = 0;
        ++g_bar_gen;
    } else {
        while (g_bar_gen == gen) yield();
    }
}
void wave_sync() {
    int w = g_cur >> 6;
    long gen = g_wave_gen[w];
    ++g_progress;
    if (++g_wave_count[w] == wave_lanes()) {
        g_wave_count[w] = 0;
        ++g_wave_gen[w];
    } else {
        while (g_wave_gen[w] == gen) yield();
    }
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    int nt = block.x * block.y * block.z;
    if (nt > kMaxThreads || nt <= 0) {
        fprintf(stderr, "emu: bad block size %d\n", nt);
        abort();
    }
    if (!g_stacks) {
        g_stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (g_stacks == MAP_FAILED) abort();
    }
    g_fibers.resize(nt);
    g_nthreads = nt;
    g_body = &body;
    gridDim = grid;
    blockDim = block;
    g_smem.assign(smem + 64, 0);
    // Workgroups run one after the other.  Real hardware promises no order, so kernels that hand data from workgroup to workgroup
    // (the last-arriver reduce of conv_wgrad.hip) are also tested in REVERSED and in an interleaved order: AVC_EMU_BLOCK_ORDER = reverse | stride
    const char* ord = getenv("AVC_EMU_BLOCK_ORDER");
    const int order = !ord ? 0 : (strcmp(ord, "reverse") == 0 ? 1 : (strcmp(ord, "stride") == 0 ? 2 : 0));
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bi = 0; bi < grid.x; ++bi) {
                unsigned bx = bi;
                if (order == 1) bx = grid.x - 1 - bi;
                if (order == 2) {   // 0, 3, 6, ..., 1, 4, 7, ..., 2, 5, 8, ...
                    const unsigned n0 = (grid.x + 2) / 3, n1 = (grid.x + 1) / 3;
                    bx = bi < n0 ? 3 * bi : (bi < n0 + n1 ? 3 * (bi - n0) + 1 : 3 * (bi - n0 - n1) + 2);
                }
                blockIdx = dim3(bx, by, bz);
                // poison dynamic LDS with NaNs so that uninitialised reads show up
                for (size_t i = 0; i + 4 <= g_smem.size(); i += 4) {
                    uint32_t nan = 0x7fc00000u;
                    memcpy(&g_smem[i], &nan, 4);
                }
                g_bar_count = 0;
                g_alive = nt;
                for (auto& c : g_wave_count) c = 0;
                for (int i = 0; i < nt; ++i) {
                    getcontext(&g_fibers[i].ctx);
                    g_fibers[i].ctx.uc_stack.ss_sp = g_stacks + kStack * i;
                    g_fibers[i].ctx.uc_stack.ss_size = kStack;
                    g_fibers[i].ctx.uc_link = nullptr;
                    g_fibers[i].done = false;
                    makecontext(&g_fibers[i].ctx, trampoline, 0);
                }
                bool alive = true;
                while (alive) {
                    alive = false;
                    long before = g_progress;
                    for (int i = 0; i < nt; ++i) {
                        if (g_fibers[i].done) continue;
                        g_cur = i;
                        set_tid(i);
                        swapcontext(&g_sched, &g_fibers[i].ctx);
                        alive |= !g_fibers[i].done;
                    }
                    if (alive && g_progress == before) {
                        fprintf(stderr, "emu: deadlock (divergent barrier?) in block (%u,%u,%u)\n", bx, by, bz);
                        abort();
                    }
                }
            }
    g_body = nullptr;
}
}  // namespace emu
