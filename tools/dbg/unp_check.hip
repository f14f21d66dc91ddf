// Debug aid: the speculative chunk tables of k_unp_spec against a host replay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mimir_gpu_shuffle.h"
#include "mgs_unpack.h"
using namespace mgs;

static uint32_t h_step(uint64_t nul, uint32_t state, uint32_t V, uint32_t valid, uint32_t &n)
{
    uint32_t pos;
    if (state & UNP_INKEY) {
        if (nul == 0) return UNP_INKEY;
        pos = (uint32_t) __builtin_ffsll((long long) nul) + V;
    }
    else pos = state;
    while (pos < valid) {
        n++;
        const uint64_t mm = nul & (~0ull << pos);
        if (mm == 0) return UNP_INKEY;
        pos = (uint32_t) __builtin_ffsll((long long) mm) + V;
    }
    return pos - valid;
}

__global__ void k_trace(UnpParams p, uint32_t c, uint32_t *trace, uint64_t *masks)
{
    const int lane = threadIdx.x;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t c1 = min(p.nbytes, c0 + p.chunk_bytes);
    const uint32_t S = p.V + 2;
    uint32_t state = ((uint32_t) lane == p.V + 1) ? UNP_INKEY : (uint32_t) lane;
    uint32_t n = 0, b = 0;
    for (uint64_t off = c0; off < c1; off += 64, b++) {
        const uint64_t nul = unp_nul_mask(p, off, lane);
        const uint32_t valid = (uint32_t) min((uint64_t) 64, c1 - off);
        uint64_t starts = 0;
        if ((uint32_t) lane < S) state = unp_step(nul, state, p.V, valid, starts, n);
        if (lane == 0) { trace[2 * b] = state; trace[2 * b + 1] = n; masks[b] = nul; }
    }
}

int main()
{
    const uint32_t V = 8, S = V + 2, chunk = 23872;
    std::vector<uint8_t> s;
    srand(5);
    std::vector<uint32_t> truth;
    while (s.size() < 15000000) {
        truth.push_back((uint32_t) s.size());
        int L = rand() % 12;
        for (int i = 0; i < L; i++) s.push_back('a' + rand() % 26);
        s.push_back(0);
        s.push_back(1);
        for (int i = 0; i < 7; i++) s.push_back(0);
    }
    const uint64_t nbytes = s.size();
    const uint32_t nchunks = (uint32_t) ((nbytes + chunk - 1) / chunk);
    uint8_t *d_s, *d_ex;
    uint32_t *d_cnt;
    hipMalloc(&d_s, nbytes);
    hipMalloc(&d_ex, nchunks * S);
    hipMalloc(&d_cnt, nchunks * S * 4);
    hipMemcpy(d_s, s.data(), nbytes, hipMemcpyHostToDevice);
    if (getenv("SYNC_AFTER_COPY")) hipDeviceSynchronize();
    UnpParams p;
    memset(&p, 0, sizeof(p));
    p.stream = d_s; p.nbytes = nbytes; p.V = V; p.chunk_bytes = chunk; p.nchunks = nchunks;
    p.exit_state = d_ex; p.count = d_cnt;
    hipLaunchKernelGGL(k_unp_spec, dim3(nchunks), dim3(64), 0, 0, p);
    hipDeviceSynchronize();
    {
        // run it again: device against device
        uint8_t *d_ex2; uint32_t *d_cnt2;
        hipMalloc(&d_ex2, nchunks * S); hipMalloc(&d_cnt2, nchunks * S * 4);
        UnpParams p2 = p; p2.exit_state = d_ex2; p2.count = d_cnt2;
        for (int rep = 0; rep < 40; rep++) {
            hipLaunchKernelGGL(k_unp_spec, dim3(nchunks), dim3(64), 0, 0, p2);
            std::vector<uint32_t> a(nchunks * S), b(nchunks * S);
            hipMemcpy(a.data(), d_cnt, a.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), d_cnt2, b.size() * 4, hipMemcpyDeviceToHost);
            int diff = 0, firstc = -1;
            for (size_t i = 0; i < a.size(); i++) if (a[i] != b[i]) { diff++; if (firstc < 0) firstc = (int) (i / S); }
            if (diff) printf("rerun %d: %d table entries differ from the first run (first chunk %d)\n", rep, diff, firstc);
        }
    }
    std::vector<uint8_t> ex(nchunks * S);
    std::vector<uint32_t> cnt(nchunks * S);
    hipMemcpy(ex.data(), d_ex, ex.size(), hipMemcpyDeviceToHost);
    hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (uint32_t c = 0; c < nchunks && bad < 10; c++) {
        for (uint32_t x = 0; x < S; x++) {
            uint32_t st = x == V + 1 ? UNP_INKEY : x, n = 0;
            for (uint64_t off = (uint64_t) c * chunk; off < std::min<uint64_t>(nbytes, (uint64_t) (c + 1) * chunk); off += 64) {
                uint64_t nul = 0;
                const uint32_t valid = (uint32_t) std::min<uint64_t>(64, nbytes - off);
                for (uint32_t i = 0; i < valid; i++) if (s[off + i] == 0) nul |= 1ull << i;
                st = h_step(nul, st, V, valid, n);
            }
            const uint32_t e = (st & UNP_INKEY) ? V + 1 : st;
            if (e != ex[c * S + x] || n != cnt[c * S + x]) {
                printf("chunk %u state %u: host exit %u count %u, device exit %u count %u\n", c, x, e, n,
                       ex[c * S + x], cnt[c * S + x]);
                bad++;
            }
        }
    }
    printf("%s (%u chunks)\n", bad ? "MISMATCH" : "tables agree", nchunks);
    if (bad) {
        uint32_t cbad = 0;
        for (uint32_t c = 0; c < nchunks; c++) {
            uint32_t st = 0, n = 0;
            for (uint64_t off = (uint64_t) c * chunk; off < std::min<uint64_t>(nbytes, (uint64_t) (c + 1) * chunk); off += 64) {
                uint64_t nul = 0;
                const uint32_t valid = (uint32_t) std::min<uint64_t>(64, nbytes - off);
                for (uint32_t i = 0; i < valid; i++) if (s[off + i] == 0) nul |= 1ull << i;
                st = h_step(nul, st, V, valid, n);
            }
            if (n != cnt[c * S]) { cbad = c; break; }
        }
        const uint32_t nb = chunk / 64;
        uint32_t *d_tr; uint64_t *d_mk;
        hipMalloc(&d_tr, nb * 8); hipMalloc(&d_mk, nb * 8);
        hipLaunchKernelGGL(k_trace, dim3(1), dim3(64), 0, 0, p, cbad, d_tr, d_mk);
        std::vector<uint32_t> tr(nb * 2); std::vector<uint64_t> mk(nb);
        hipMemcpy(tr.data(), d_tr, nb * 8, hipMemcpyDeviceToHost);
        hipMemcpy(mk.data(), d_mk, nb * 8, hipMemcpyDeviceToHost);
        uint32_t st = 0, n = 0;
        for (uint32_t b = 0; b < nb; b++) {
            const uint64_t off = (uint64_t) cbad * chunk + (uint64_t) b * 64;
            uint64_t nul = 0;
            const uint32_t valid = (uint32_t) std::min<uint64_t>(64, nbytes - off);
            for (uint32_t i = 0; i < valid; i++) if (s[off + i] == 0) nul |= 1ull << i;
            const uint32_t st_in = st;
            st = h_step(nul, st, V, valid, n);
            if (st != tr[2 * b] || n != tr[2 * b + 1] || nul != mk[b]) {
                printf("chunk %u block %u: in-state %x host (state %x n %u nul %016llx) device (state %x n %u nul %016llx)\n",
                       cbad, b, st_in, st, n, (unsigned long long) nul, tr[2 * b], tr[2 * b + 1], (unsigned long long) mk[b]);
                break;
            }
        }
    }
    truth.push_back((uint32_t) nbytes);
    // chain + emit
    uint32_t *d_entry, *d_off;
    uint64_t *d_first, *d_res;
    Ctrl *d_ctrl;
    hipMalloc(&d_entry, nchunks * 4);
    hipMalloc(&d_first, (nchunks + 1) * 8);
    hipMalloc(&d_res, 16);
    hipMalloc(&d_off, truth.size() * 4 + 64);
    hipMalloc(&d_ctrl, sizeof(Ctrl));
    hipMemset(d_ctrl, 0, sizeof(Ctrl));
    hipMemset(d_off, 0xff, truth.size() * 4 + 64);
    p.entry = d_entry; p.first_rec = d_first; p.result = d_res; p.rec_off = d_off; p.rec_cap = truth.size() + 8;
    p.ctrl = d_ctrl;
    hipLaunchKernelGGL(k_unp_chain, dim3(1), dim3(256), (size_t) nchunks * S * 5, 0, p);
    hipLaunchKernelGGL(k_unp_emit, dim3(nchunks), dim3(64), 0, 0, p);
    std::vector<uint32_t> off(truth.size());
    std::vector<uint64_t> first(nchunks + 1);
    std::vector<uint32_t> entry(nchunks);
    hipMemcpy(off.data(), d_off, off.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(first.data(), d_first, first.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(entry.data(), d_entry, entry.size() * 4, hipMemcpyDeviceToHost);
    printf("records: truth %zu device %llu, err %s\n", truth.size() - 1, (unsigned long long) first[nchunks], hipGetErrorString(hipGetLastError()));
    // host chain
    uint32_t st = 0; uint64_t total = 0;
    for (uint32_t c = 0; c < nchunks; c++) {
        if (entry[c] != st || first[c] != total) { printf("chain differs at chunk %u: entry %u/%u first %llu/%llu\n", c, entry[c], st, (unsigned long long) first[c], (unsigned long long) total); break; }
        total += cnt[c * S + st]; st = ex[c * S + st];
    }
    for (size_t i = 0; i < truth.size(); i++) if (off[i] != truth[i]) { printf("offset %zu: device %u truth %u (chunk %u)\n", i, off[i], truth[i], truth[i] / chunk); bad++; if (bad > 5) break; }
    return bad != 0;
}
