// hcp_emu.cpp — scheduler of the test-only wave64 interpreter (see hcp_emu.h).
#include "hcp_emu.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" void hcp_emu_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl hcp_emu_ctx_switch
.type hcp_emu_ctx_switch,@function
hcp_emu_ctx_switch:
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
)");

namespace hcp_emu {
Fiber* g_cur = nullptr;
uint3_t g_block = {0, 0, 0};
dim3 g_bdim, g_gdim;
unsigned char* g_smem = nullptr;

static void* g_main_sp = nullptr;
static const std::function<void()>* g_body = nullptr;
static const size_t kStack = 256 * 1024;
static const size_t kSmem = 160 * 1024;

static float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

static void to_main() { hcp_emu_ctx_switch(&g_cur->sp, g_main_sp); }
void yield_barrier() { g_cur->state = ST_BARRIER; to_main(); }
void wave_collective() { g_cur->state = ST_WAVE; to_main(); }

int g_dma_deferred = 0;
static const int kPend = 1024;
void dma_push(void* dst, const void* src, int n) {
    if (!g_dma_deferred) { if (src) memcpy(dst, src, n); else memset(dst, 0, n); return; }
    Fiber* f = g_cur;
    if (f->pend_n >= kPend) { fprintf(stderr, "hcp_emu: more than %d LDS-DMA copies in flight in one lane\n", kPend); abort(); }
    Fiber::Pend& e = f->pend[f->pend_n++];
    e.dst = (unsigned char*)dst; e.n = n;
    if (src) memcpy(e.data, src, n); else memset(e.data, 0, n);      // the source is read at issue: kernel inputs do not change under a launch
}
void dma_drain(int keep) {
    Fiber* f = g_cur;
    if (f->pend_n <= keep) return;
    const int land = f->pend_n - (keep > 0 ? keep : 0);
    for (int i = 0; i < land; ++i) memcpy(f->pend[i].dst, f->pend[i].data, f->pend[i].n);      // loads return in order: the oldest land first
    memmove(f->pend, f->pend + land, sizeof(Fiber::Pend) * (f->pend_n - land));
    f->pend_n -= land;
}

static void trampoline() {
    (*g_body)();
    dma_drain(0);
    g_cur->state = ST_DONE;
    to_main();
    fprintf(stderr, "hcp_emu: resumed a finished fibre\n");
    abort();
}

static void init_fiber(Fiber& f) {
    char* top = f.stack + kStack;
    top = (char*)((uintptr_t)top & ~(uintptr_t)15);
    void** base = (void**)(top - 64);
    for (int i = 0; i < 6; ++i) base[i] = nullptr;
    base[6] = (void*)&trampoline;
    base[7] = nullptr;
    f.sp = base;
}

static void resolve_wave(std::vector<Fiber>& fb, int w0, int w1) {
    int op = 0;
    for (int i = w0; i < w1; ++i)
        if (fb[i].state == ST_WAVE) {
            if (op == 0) op = fb[i].op;
            else if (op != fb[i].op) { fprintf(stderr, "hcp_emu: divergent wave collective (%d vs %d)\n", op, fb[i].op); abort(); }
        }
    auto in = [&](int lane, int word) -> uint32_t {
        int i = w0 + lane;
        if (i >= w1 || fb[i].state != ST_WAVE) return 0u;
        return fb[i].in[word];
    };
    auto bfe = [&](int lane, int base, int j) -> float {  // element j of an 8 x bf16 operand
        uint32_t w = in(lane, base + (j >> 1));
        return bf2f((uint16_t)((j & 1) ? (w >> 16) : (w & 0xffff)));
    };
    if (op == OP_SHFL) {
        for (int i = w0; i < w1; ++i)
            if (fb[i].state == ST_WAVE) fb[i].out[0] = in(fb[i].src_lane, 0);
    } else if (op == OP_TR16) {
        // lane i of a 16-lane group receives { M[4j + (i>>2)][i&3] : j }, M[p] = the 4 bf16 lane p pointed at
        for (int l = 0; l < 64 && w0 + l < w1; ++l) {
            if (fb[w0 + l].state != ST_WAVE) continue;
            const int base = l & ~15, i = l & 15;
            uint16_t o[4];
            for (int j = 0; j < 4; ++j) {
                const int src = base + 4 * j + (i >> 2), e = i & 3;
                uint32_t w = in(src, e >> 1);
                o[j] = (uint16_t)((e & 1) ? (w >> 16) : (w & 0xffff));
            }
            memcpy(fb[w0 + l].out, o, 8);
        }
    } else if (op == OP_MFMA16) {
        // A[i][k]: lane = i + 16*(k/8), elem k%8 ; B[k][j]: lane = j + 16*(k/8), elem k%8
        // D[i][j]: lane = j + 16*(i/4), reg i%4
        for (int l = 0; l < 64 && w0 + l < w1; ++l) {
            if (fb[w0 + l].state != ST_WAVE) continue;
            int j = l & 15;
            for (int r = 0; r < 4; ++r) {
                int i = (l >> 4) * 4 + r;
                float acc; uint32_t cw = in(l, 8 + r); memcpy(&acc, &cw, 4);
                for (int k = 0; k < 32; ++k)
                    acc += bfe(i + 16 * (k >> 3), 0, k & 7) * bfe(j + 16 * (k >> 3), 4, k & 7);
                memcpy(&fb[w0 + l].out[r], &acc, 4);
            }
        }
    } else if (op == OP_MFMA32) {
        // A[i][k]: lane = i + 32*(k/8) ; B[k][j]: lane = j + 32*(k/8) ; D[i][j]: lane = j + 32*((i/4)%2),
        // reg = (i%4) + 4*(i/8)
        for (int l = 0; l < 64 && w0 + l < w1; ++l) {
            if (fb[w0 + l].state != ST_WAVE) continue;
            int j = l & 31;
            for (int r = 0; r < 16; ++r) {
                int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                float acc; uint32_t cw = in(l, 8 + r); memcpy(&acc, &cw, 4);
                for (int k = 0; k < 16; ++k)
                    acc += bfe(i + 32 * (k >> 3), 0, k & 7) * bfe(j + 32 * (k >> 3), 4, k & 7);
                memcpy(&fb[w0 + l].out[r], &acc, 4);
            }
        }
    } else {
        fprintf(stderr, "hcp_emu: unknown collective %d\n", op); abort();
    }
    for (int i = w0; i < w1; ++i)
        if (fb[i].state == ST_WAVE) fb[i].state = ST_RUN;
}

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body) {
    static std::vector<Fiber> fb;
    static unsigned char* smem_buf = nullptr;
    if (!smem_buf) smem_buf = (unsigned char*)aligned_alloc(64, kSmem);
    if (smem > kSmem) { fprintf(stderr, "hcp_emu: LDS request %zu > 160 KiB\n", smem); abort(); }
    const int n = (int)(block.x * block.y * block.z);
    if (n > 1024 || n <= 0) { fprintf(stderr, "hcp_emu: bad block size %d\n", n); abort(); }
    while ((int)fb.size() < n) {
        Fiber f; memset(&f, 0, sizeof(f));
        f.stack = (char*)aligned_alloc(64, kStack);
        f.pend = (Fiber::Pend*)malloc(sizeof(Fiber::Pend) * kPend);
        fb.push_back(f);
    }
    g_bdim = block; g_gdim = grid; g_smem = smem_buf; g_body = &body;
    for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_block = {bx, by, bz};
        for (int t = 0; t < n; ++t) {
            Fiber& f = fb[t];
            f.lin = t; f.wave = t >> 6; f.lane = t & 63; f.state = ST_RUN; f.pend_n = 0;
            f.tid.x = t % block.x; f.tid.y = (t / block.x) % block.y; f.tid.z = t / (block.x * block.y);
            init_fiber(f);
        }
        int done = 0;
        while (done < n) {
            bool progress = false;
            for (int t = 0; t < n; ++t) {
                if (fb[t].state != ST_RUN) continue;
                g_cur = &fb[t];
                hcp_emu_ctx_switch(&g_main_sp, fb[t].sp);
                progress = true;
                if (fb[t].state == ST_DONE) ++done;
            }
            bool released = false;
            for (int w0 = 0; w0 < n; w0 += 64) {
                int w1 = w0 + 64 < n ? w0 + 64 : n;
                int nw = 0, nother = 0;
                for (int i = w0; i < w1; ++i) {
                    if (fb[i].state == ST_WAVE) ++nw;
                    else if (fb[i].state != ST_DONE) ++nother;
                }
                if (nw > 0 && nother == 0) { resolve_wave(fb, w0, w1); released = true; }
            }
            int nb = 0, nlive = 0;
            for (int t = 0; t < n; ++t) {
                if (fb[t].state == ST_BARRIER) ++nb;
                if (fb[t].state != ST_DONE) ++nlive;
            }
            if (nlive > 0 && nb == nlive) {
                for (int t = 0; t < n; ++t) if (fb[t].state == ST_BARRIER) fb[t].state = ST_RUN;
                released = true;
            }
            if (!progress && !released && done < n) {
                fprintf(stderr, "hcp_emu: deadlock in block (%u,%u,%u): %d live, %d at barrier\n", bx, by, bz, nlive, nb);
                abort();
            }
        }
    }
    g_cur = nullptr;
}
}  // namespace hcp_emu

// TEST ONLY: 1 = LDS-DMA copies land at the wait that retires them (latest the hardware allows), 0 = at issue (earliest).
extern "C" __attribute__((visibility("default"))) int hcp_debug_emu_dma_deferred(int on) { hcp_emu::g_dma_deferred = on ? 1 : 0; return 0; }
