/* simt_emu.cpp -- TEST INFRASTRUCTURE ONLY (see simt_emu.h): fiber scheduler. */
#include "simt_emu.h"
#include <mutex>
#include <algorithm>

namespace simt {

thread_local Block* g_blk = nullptr;

// void simt_switch(void** save_sp, void* load_sp): save callee-saved regs + sp, load other
__asm__(
    ".text\n"
    ".globl simt_switch\n"
    ".type simt_switch,@function\n"
    "simt_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size simt_switch,.-simt_switch\n");

static void fiber_entry() {
    Block* b = g_blk;
    Fiber* f = b->cur;
    (*b->body)();
    f->done = true;
    f->wave->live--;
    b->live_threads--;
    b->progress++;
    for (;;) yield_to_sched();
}

static void run_block(Block& b, size_t stack_bytes, char* stacks) {
    g_blk = &b;
    const unsigned nt = b.nthreads, nw = nt / 64;
    b.fibers.assign(nt, Fiber{});
    b.waves.assign(nw, Wave{});
    b.bar = Rendezvous{};
    b.live_threads = nt;
    b.progress = 0;
    for (unsigned w = 0; w < nw; w++) { b.waves[w].live = 64; b.waves[w].first = w * 64; b.waves[w].rv = Rendezvous{}; }
    for (unsigned t = 0; t < nt; t++) {
        Fiber& f = b.fibers[t];
        f.tid = t; f.done = false; f.gen = 0; f.bar_gen = 0;
        f.wave = &b.waves[t / 64]; f.block = &b;
        f.stack = stacks + (size_t)t * stack_bytes;
        uintptr_t top = ((uintptr_t)f.stack + stack_bytes) & ~(uintptr_t)15;
        void** sp = (void**)top;
        *--sp = nullptr;                    // fake return address of fiber_entry
        *--sp = (void*)&fiber_entry;        // `ret` target of the first switch
        for (int i = 0; i < 6; i++) *--sp = nullptr;   // rbp rbx r12-r15
        f.sp = sp;
    }
    // scheduler: wave by wave, several rounds per wave while it makes progress.  SIMT_SEED=<n>: the waves of a pass in a pseudo-random
    // order, a pseudo-random number of rounds each - protocols between waves must not depend on who runs first
    unsigned long long last_progress = ~0ull;
    unsigned idle_passes = 0;
    static const char* seed_env = getenv("SIMT_SEED");
    unsigned long long rng = seed_env ? 0x9E3779B97F4A7C15ull * (unsigned long long)(atoll(seed_env) + 1) + b.bIdx.x : 0;
    std::vector<unsigned> order(nw);
    for (unsigned w = 0; w < nw; w++) order[w] = w;
    while (b.live_threads) {
        last_progress = b.progress;
        int max_rounds = 16;
        if (seed_env) {
            for (unsigned i = nw; i > 1; i--) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; std::swap(order[i - 1], order[(rng >> 33) % i]); }
            rng = rng * 6364136223846793005ull + 1442695040888963407ull; max_rounds = 1 + (int)((rng >> 33) % 16);
        }
        for (unsigned wi = 0; wi < nw && b.live_threads; wi++) {
            const unsigned w = order[wi];
            if (!b.waves[w].live) continue;
            for (int round = 0; round < max_rounds; round++) {
                unsigned long long before = b.progress;
                for (unsigned l = 0; l < 64; l++) {
                    Fiber& f = b.fibers[w * 64 + l];
                    if (f.done) continue;
                    b.cur = &f;
                    simt_switch(&b.sched_sp, f.sp);
                }
                if (b.progress == before || !b.waves[w].live) break;
            }
        }
        if (b.progress == last_progress) {
            if (++idle_passes > 4) {
                fprintf(stderr, "simt: barrier count %u/%u gen %u/%u\n", b.bar.count[0], b.bar.count[1], b.bar.gen[0], b.bar.gen[1]);
                fprintf(stderr, "simt: DEADLOCK in block %u (live threads %u): a wave-wide builtin or "
                        "__syncthreads() was not reached by every live lane\n", b.bIdx.x, b.live_threads);
                for (unsigned w = 0; w < nw; w++)
                    if (b.waves[w].live) {
                        unsigned g0 = ~0u, g1 = 0, b0 = ~0u, b1 = 0;
                        for (unsigned l = 0; l < 64; l++) { Fiber& f = b.fibers[w * 64 + l]; if (f.done) continue;
                            if (f.gen < g0) g0 = f.gen; if (f.gen > g1) g1 = f.gen;
                            if (f.bar_gen < b0) b0 = f.bar_gen; if (f.bar_gen > b1) b1 = f.bar_gen; }
                        unsigned nwv = 0, nbar = 0;
                        for (unsigned l = 0; l < 64; l++) { Fiber& f = b.fibers[w * 64 + l]; if (f.done) continue; nwv += f.waiting == 1; nbar += f.waiting == 2; }
                        fprintf(stderr, "  wave %u: live %u  wave-op gen %u..%u  barrier gen %u..%u  in-wave-op %u in-barrier %u  rv.count %u/%u\n", w, b.waves[w].live, g0, g1, b0, b1, nwv, nbar, b.waves[w].rv.count[0], b.waves[w].rv.count[1]);
                    }
                abort();
            }
        } else idle_passes = 0;
    }
    g_blk = nullptr;
}

void launch(unsigned grid, unsigned block, size_t smem_bytes, const std::function<void()>& body,
            size_t stack_bytes) {
    if (block % 64 || block == 0) { fprintf(stderr, "simt: blockDim must be a multiple of 64\n"); abort(); }
    unsigned nthr = 1;
    if (const char* e = getenv("SIMT_THREADS")) nthr = (unsigned)atoi(e);
    else { nthr = std::thread::hardware_concurrency(); if (!nthr) nthr = 1; }
    if (nthr > grid) nthr = grid;
    if (nthr < 1) nthr = 1;
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        char* stacks = (char*)malloc((size_t)block * stack_bytes + 64);
        char* smem = (char*)malloc(smem_bytes + 64);
        Block b;
        for (;;) {
            unsigned bi = next.fetch_add(1);
            if (bi >= grid) break;
            memset(smem, 0xCD, smem_bytes + 64);      // LDS is garbage at block start
            b.nthreads = block;
            b.smem = (char*)(((uintptr_t)smem + 15) & ~(uintptr_t)15);
            b.bIdx = {bi, 0, 0}; b.bDim = {block, 1, 1}; b.gDim = {grid, 1, 1};
            b.body = &body;
            run_block(b, stack_bytes, stacks);
        }
        free(stacks); free(smem);
    };
    if (nthr == 1) worker();
    else {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nthr; i++) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
}

} // namespace simt
