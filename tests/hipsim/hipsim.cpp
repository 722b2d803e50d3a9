// hipsim.cpp - fiber-based workgroup executor behind hipsim.h (test infrastructure only).
#include "hipsim.h"

#include <sys/mman.h>

#include <mutex>
#include <vector>

namespace sim {
thread_local Lane* cur = nullptr;
thread_local Block* curblk = nullptr;

static constexpr size_t STACK_BYTES = 256 * 1024;

// void sim_switch(Ctx* from, Ctx* to): push the SysV callee-saved registers, swap stack pointers, pop, return.
asm(R"(
    .text
    .globl sim_switch
    .type sim_switch,@function
sim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size sim_switch, .-sim_switch
)");

// Fiber stacks are recycled between launches (a fresh mmap per launch paid its page faults again every time).
static std::mutex stack_mu;
static std::vector<void*> stack_pool;
static void* stack_get() {
    {
        std::lock_guard<std::mutex> g(stack_mu);
        if (!stack_pool.empty()) { void* s = stack_pool.back(); stack_pool.pop_back(); return s; }
    }
    void* s = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
    if (s == MAP_FAILED) { std::perror("mmap"); std::abort(); }
    return s;
}
static void stack_put(std::vector<void*>& v) {
    std::lock_guard<std::mutex> g(stack_mu);
    for (void* s : v) stack_pool.push_back(s);
    v.clear();
}

static void lane_entry() {
    Lane* l = cur;
    (*l->blk->body)();
    for (const Lane::PendingDma& d : l->dma)         // (late-DMA mode: whatever a kernel left in flight)
        if (d.bytes) std::memcpy(d.dst, d.src, (size_t)d.bytes);
    l->dma.clear();
    l->done = true;
    Block* b = l->blk;
    WaveState& w = b->waves[l->wave];
    // an exiting lane no longer takes part in barriers; release a barrier it was the last missing arrival of
    --w.alive;
    if (w.alive > 0 && w.arrived >= w.alive) { w.arrived = 0; ++w.gen; }
    --b->alive;
    if (b->alive > 0 && b->arrived >= b->alive) { b->arrived = 0; ++b->gen; }
    sim_switch(&l->ctx, &b->sched);
    std::abort();                                    // a finished lane is never resumed
}

static void run_block(Block& b, std::vector<void*>& stacks) {
    const int nthreads = b.bdim.x * b.bdim.y * b.bdim.z;
    const int nwaves = (nthreads + WAVE - 1) / WAVE;
    b.lanes.assign(nthreads, Lane());
    b.waves.assign(nwaves, WaveState());
    b.alive = nthreads;
    b.arrived = 0;
    b.gen = 0;
    for (int t = 0; t < nthreads; ++t) {
        Lane& l = b.lanes[t];
        l.linear = t;
        l.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
        l.wave = t / WAVE;
        l.lane = t % WAVE;
        l.blk = &b;
        l.stack = stacks[t];
        b.waves[l.wave].alive++;
        // initial frame: six zeroed callee-saved registers, then lane_entry as the "return address"; after that `ret`
        // rsp == top - 8, i.e. the alignment a function sees right after a call
        uintptr_t top = (reinterpret_cast<uintptr_t>(l.stack) + STACK_BYTES) & ~uintptr_t(15);
        void** frame = reinterpret_cast<void**>(top - 64);
        for (int i = 0; i < 6; ++i) frame[i] = nullptr;
        frame[6] = reinterpret_cast<void*>(&lane_entry);
        frame[7] = nullptr;
        l.ctx.rsp = frame;
    }
    curblk = &b;
    int remaining = nthreads;
    while (remaining > 0) {
        bool progressed = false;
        for (int t = 0; t < nthreads; ++t) {
            Lane& l = b.lanes[t];
            if (l.done) continue;
            if (l.wait_kind == 1 && b.waves[l.wave].gen == l.wait_gen) continue;
            if (l.wait_kind == 2 && b.gen == l.wait_gen) continue;
            l.wait_kind = 0;
            cur = &l;
            sim_switch(&b.sched, &l.ctx);
            progressed = true;
            if (l.done) --remaining;
        }
        if (!progressed) {
            std::fprintf(stderr, "hipsim: deadlock in block (%u,%u,%u): barrier not reached by all lanes\n", b.bid.x,
                         b.bid.y, b.bid.z);
            std::abort();
        }
    }
    cur = nullptr;
    curblk = nullptr;
}

void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body) {
    const size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    const int nthreads = block.x * block.y * block.z;
    unsigned hw = std::thread::hardware_concurrency();
    const size_t nworkers = std::min<size_t>(nblocks, hw ? hw : 4);
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        std::vector<void*> stacks(nthreads);
        for (auto& s : stacks) s = stack_get();
        std::vector<char> smem(dyn_smem + 64);
        Block b;
        b.bdim = block;
        b.gdim = grid;
        b.body = &body;
        b.dyn_smem = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(smem.data()) + 63) & ~uintptr_t(63));
        for (;;) {
            size_t i = next.fetch_add(1);
            if (i >= nblocks) break;
            b.bid = dim3(i % grid.x, (i / grid.x) % grid.y, i / ((size_t)grid.x * grid.y));
            run_block(b, stacks);
        }
        stack_put(stacks);
    };
    std::vector<std::thread> pool;
    for (size_t w = 1; w < nworkers; ++w) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
}
}  // namespace sim
