// Runtime half of the test-only HIP stand-in (see include/hip/hip_runtime.h).
//
// Execution model: the work-items of a workgroup are FIBERS (user-space contexts with their own stacks) of the host thread
// that launches the kernel; workgroups run one after another; a launch holds a process-wide lock (`__shared__` is
// function-scope static storage, one copy).  A fiber runs until it reaches a barrier -- __syncthreads(), the wave-local
// rendezvous behind wave_sync() / the cross-lane builtins -- or returns; the scheduler then resumes the next runnable one.
// A barrier releases when every work-item of its group (workgroup / wavefront) that has not yet RETURNED has arrived, which
// is the hardware's rule for ended wavefronts.  No system call anywhere on that path: the first form of this runtime (an OS
// thread per work-item on pthread barriers) spent 90 % of the CPU suite's time in futex calls.
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <mutex>

#if !defined(__x86_64__)
#include <ucontext.h>
#endif

namespace emu {
thread_local Idx t_threadIdx, t_blockIdx;
Idx g_blockDim, g_gridDim;
uint32_t g_exchange[1024];

namespace {

constexpr size_t kStackBytes = 512u << 10;  // per work-item (virtual; touched pages only)
enum State : uint8_t { RUNNABLE, WAIT_WAVE, WAIT_WG, DONE };

struct Fiber {
#if defined(__x86_64__)
    void *sp = nullptr;
#else
    ucontext_t uc;
#endif
    char *stack = nullptr;
    State state = DONE;
};

std::mutex g_launch_lock;
std::vector<Fiber> g_fibers;  // grown on demand, stacks kept between launches
#if defined(__x86_64__)
void *g_sched_sp = nullptr;
#else
ucontext_t g_sched_uc;
#endif
unsigned g_cur = 0, g_n = 0;
const std::function<void()> *g_body = nullptr;
unsigned g_wave_arrived[32], g_wave_live[32], g_wg_arrived, g_wg_live;

#if defined(__x86_64__)
// save the callee-saved registers and the stack pointer of the running context in *save, continue on `load`
extern "C" void emu_switch(void **save, void *load);
asm(R"(
    .text
    .globl emu_switch
    .hidden emu_switch
    .type emu_switch, @function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size emu_switch, .-emu_switch
)");
inline void to_scheduler(Fiber &f) { emu_switch(&f.sp, g_sched_sp); }
inline void to_fiber(Fiber &f) { emu_switch(&g_sched_sp, f.sp); }
#else
inline void to_scheduler(Fiber &f) { swapcontext(&f.uc, &g_sched_uc); }
inline void to_fiber(Fiber &f) { swapcontext(&g_sched_uc, &f.uc); }
#endif

void release_wave(unsigned w) {
    g_wave_arrived[w] = 0;
    const unsigned lo = w * 64, hi = lo + 64 < g_n ? lo + 64 : g_n;
    for (unsigned i = lo; i < hi; ++i)
        if (g_fibers[i].state == WAIT_WAVE) g_fibers[i].state = RUNNABLE;
}
void release_wg() {
    g_wg_arrived = 0;
    for (unsigned i = 0; i < g_n; ++i)
        if (g_fibers[i].state == WAIT_WG) g_fibers[i].state = RUNNABLE;
}

void fiber_main() {
    (*g_body)();
    // this work-item has returned: it no longer counts at any barrier, which may complete one
    Fiber &f = g_fibers[g_cur];
    f.state = DONE;
    const unsigned w = g_cur >> 6;
    if (--g_wave_live[w] > 0 && g_wave_arrived[w] == g_wave_live[w]) release_wave(w);
    if (--g_wg_live > 0 && g_wg_arrived == g_wg_live) release_wg();
    to_scheduler(f);
    abort();  // (a finished fiber is never resumed)
}

void prepare(Fiber &f) {
    if (!f.stack) {
        void *m = mmap(nullptr, kStackBytes + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) {
            fprintf(stderr, "emu: no memory for a work-item stack\n");
            abort();
        }
        mprotect(m, 4096, PROT_NONE);  // guard page below the stack
        f.stack = static_cast<char *>(m) + 4096;
    }
    f.state = RUNNABLE;
#if defined(__x86_64__)
    // the frame emu_switch pops: six registers, then `ret` into fiber_main with the stack aligned as after a call
    void **top = reinterpret_cast<void **>(f.stack + kStackBytes);
    top[-1] = nullptr;                                  // (fiber_main's return address: never used)
    top[-2] = reinterpret_cast<void *>(&fiber_main);
    for (int i = 3; i <= 8; ++i) top[-i] = nullptr;
    f.sp = top - 8;
#else
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = f.stack;
    f.uc.uc_stack.ss_size = kStackBytes;
    f.uc.uc_link = nullptr;
    makecontext(&f.uc, fiber_main, 0);
#endif
}

void run_block(dim3 block) {
    for (unsigned i = 0; i < g_n; ++i) prepare(g_fibers[i]);
    const unsigned nwaves = (g_n + 63) / 64;
    for (unsigned w = 0; w < nwaves; ++w) {
        g_wave_arrived[w] = 0;
        g_wave_live[w] = (w + 1) * 64 <= g_n ? 64 : g_n - w * 64;
    }
    g_wg_arrived = 0;
    g_wg_live = g_n;
    unsigned remaining = g_n, next = 0;
    while (remaining) {
        unsigned scanned = 0;
        while (g_fibers[next].state != RUNNABLE) {
            next = next + 1 == g_n ? 0 : next + 1;
            if (++scanned > g_n) {
                fprintf(stderr, "emu: barrier deadlock -- no runnable work-item (%u of %u still live)\n", remaining, g_n);
                abort();
            }
        }
        g_cur = next;
        t_threadIdx = Idx{next % block.x, (next / block.x) % block.y, next / (block.x * block.y)};
        to_fiber(g_fibers[next]);
        if (g_fibers[next].state == DONE) --remaining;
        if (g_fibers[next].state != RUNNABLE) next = next + 1 == g_n ? 0 : next + 1;
    }
}

}  // namespace

void wave_barrier() {
    Fiber &f = g_fibers[g_cur];
    const unsigned w = g_cur >> 6;
    f.state = WAIT_WAVE;
    if (++g_wave_arrived[w] == g_wave_live[w]) release_wave(w);
    if (f.state != RUNNABLE) to_scheduler(f);
}

void workgroup_barrier() {
    Fiber &f = g_fibers[g_cur];
    f.state = WAIT_WG;
    if (++g_wg_arrived == g_wg_live) release_wg();
    if (f.state != RUNNABLE) to_scheduler(f);
}

void launch(dim3 grid, dim3 block, const std::function<void()> &body) {
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned nblocks = grid.x * grid.y * grid.z;
    if (nthreads == 0 || nblocks == 0) return;
    if (nthreads > 1024) {
        fprintf(stderr, "emu: %u work-items per workgroup\n", nthreads);
        abort();
    }
    std::lock_guard<std::mutex> hold(g_launch_lock);
    const Idx caller_thread = t_threadIdx, caller_block = t_blockIdx;
    g_blockDim = Idx{block.x, block.y, block.z};
    g_gridDim = Idx{grid.x, grid.y, grid.z};
    if (g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    g_n = nthreads;
    g_body = &body;
    for (unsigned b = 0; b < nblocks; b++) {
        t_blockIdx = Idx{b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y)};
        run_block(block);
    }
    g_body = nullptr;
    t_threadIdx = caller_thread;
    t_blockIdx = caller_block;
}
}  // namespace emu
