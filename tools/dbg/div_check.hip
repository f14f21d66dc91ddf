// Debug aid: is a lane-divergent parse loop deterministic on gfx950?
// Lane x of each wave parses the wave's chunk from entry state x, every lane
// walking the same 64-bit NUL masks with its own position.  Variant 0 leaves the
// loop with an early return, variant 1 has a single loop exit.  Both are run
// repeatedly and compared with a host replay.  Usage: div_check [first variant] [reps]
// Seen so far: both variants agree with the host on every launch but, now and
// then, the very first launch of a process on a fresh box, where one chunk's
// counts are those of a stretch of zero bytes.  To tell stale input from a wrong
// kernel, every launch also leaves a checksum of the bytes each chunk's wave
// actually read; on a mismatch the chunk is copied back from the device and
// compared with the host bytes, and its checksum as seen by the kernel is printed
// beside the host's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr uint32_t INKEY = 0x80000000u;

template <int VARIANT>
__device__ __forceinline__ uint32_t step(uint64_t nul, uint32_t state, uint32_t V, uint32_t valid, uint32_t &n)
{
    uint32_t pos;
    if (state & INKEY) {
        if (nul == 0) return INKEY;
        pos = (uint32_t) __ffsll((unsigned long long) nul) + V;
    }
    else pos = state;
    if (VARIANT == 0) {
        while (pos < valid) {
            n++;
            const uint64_t mm = nul & (~0ull << pos);
            if (mm == 0) return INKEY;
            pos = (uint32_t) __ffsll((unsigned long long) mm) + V;
        }
        return pos - valid;
    }
    bool inkey = false;
    while (pos < valid && !inkey) {
        n++;
        const uint64_t mm = nul & (~0ull << pos);
        if (mm == 0) inkey = true;
        else pos = (uint32_t) __ffsll((unsigned long long) mm) + V;
    }
    return inkey ? INKEY : pos - valid;
}

template <int VARIANT>
__global__ __launch_bounds__(64) void k_div(const uint8_t *s, uint64_t nbytes, uint32_t V, uint32_t chunk,
                                            uint32_t *exit_state, uint32_t *count, uint32_t *csum)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x, S = V + 2;
    const uint64_t c0 = (uint64_t) c * chunk, c1 = min(nbytes, c0 + chunk);
    uint32_t state = ((uint32_t) lane == V + 1) ? INKEY : (uint32_t) lane, n = 0, sum = 0;
    for (uint64_t off = c0; off < c1; off += 64) {
        const uint64_t i = off + lane;
        const uint32_t byte = (i < c1) ? s[i] : 0u;
        sum += byte * (uint32_t) (1 + (i & 0xffu));
        const uint64_t nul = __ballot(i < nbytes && byte == 0 && i < c1);
        const uint32_t valid = (uint32_t) min((uint64_t) 64, c1 - off);
        if ((uint32_t) lane < S) state = step<VARIANT>(nul, state, V, valid, n);
    }
    if ((uint32_t) lane < S) {
        exit_state[c * S + lane] = state;
        count[c * S + lane] = n;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor((int) sum, o);
    if (lane == 0) csum[c] = sum;
}

static uint32_t h_step(uint64_t nul, uint32_t state, uint32_t V, uint32_t valid, uint32_t &n)
{
    uint32_t pos;
    if (state & INKEY) {
        if (nul == 0) return INKEY;
        pos = (uint32_t) __builtin_ffsll((long long) nul) + V;
    }
    else pos = state;
    while (pos < valid) {
        n++;
        const uint64_t mm = nul & (~0ull << pos);
        if (mm == 0) return INKEY;
        pos = (uint32_t) __builtin_ffsll((long long) mm) + V;
    }
    return pos - valid;
}

int main(int argc, char **argv)
{
    const int first = argc > 1 ? atoi(argv[1]) : 0, reps = argc > 2 ? atoi(argv[2]) : 400;
    const uint32_t V = 8, S = V + 2, chunk = 23872;
    std::vector<uint8_t> s;
    srand(5);
    while (s.size() < 15000000) {
        int L = rand() % 12;
        for (int i = 0; i < L; i++) s.push_back('a' + rand() % 26);
        s.push_back(0);
        s.push_back(1);
        for (int i = 0; i < 7; i++) s.push_back(0);
    }
    const uint64_t nbytes = s.size();
    const uint32_t nchunks = (uint32_t) ((nbytes + chunk - 1) / chunk);
    std::vector<uint32_t> hex(nchunks * S), hcnt(nchunks * S);
    for (uint32_t c = 0; c < nchunks; c++)
        for (uint32_t x = 0; x < S; x++) {
            uint32_t st = x == V + 1 ? INKEY : x, n = 0;
            for (uint64_t off = (uint64_t) c * chunk; off < std::min<uint64_t>(nbytes, (uint64_t) (c + 1) * chunk); off += 64) {
                uint64_t nul = 0;
                const uint32_t valid = (uint32_t) std::min<uint64_t>(64, nbytes - off);
                for (uint32_t i = 0; i < valid; i++) if (s[off + i] == 0) nul |= 1ull << i;
                st = h_step(nul, st, V, valid, n);
            }
            hex[c * S + x] = st;
            hcnt[c * S + x] = n;
        }
    std::vector<uint32_t> hsum(nchunks), dsum(nchunks);
    for (uint32_t c = 0; c < nchunks; c++) {
        uint32_t sum = 0;
        for (uint64_t i = (uint64_t) c * chunk; i < std::min<uint64_t>(nbytes, (uint64_t) (c + 1) * chunk); i++)
            sum += (uint32_t) s[i] * (uint32_t) (1 + (i & 0xffu));
        hsum[c] = sum;
    }
    uint8_t *d_s;
    uint32_t *d_ex, *d_cnt, *d_sum;
    hipMalloc(&d_sum, nchunks * 4);
    hipMalloc(&d_s, nbytes);
    hipMalloc(&d_ex, nchunks * S * 4);
    hipMalloc(&d_cnt, nchunks * S * 4);
    hipMemcpy(d_s, s.data(), nbytes, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    std::vector<uint32_t> ex(nchunks * S), cnt(nchunks * S);
    for (int vi = 0; vi < 2; vi++) {
        const int variant = vi ^ first;
        int bad_runs = 0, bad_entries = 0;
        for (int rep = 0; rep < reps; rep++) {
            if (variant == 0) hipLaunchKernelGGL(k_div<0>, dim3(nchunks), dim3(64), 0, 0, d_s, nbytes, V, chunk, d_ex, d_cnt, d_sum);
            else hipLaunchKernelGGL(k_div<1>, dim3(nchunks), dim3(64), 0, 0, d_s, nbytes, V, chunk, d_ex, d_cnt, d_sum);
            hipMemcpy(ex.data(), d_ex, ex.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(cnt.data(), d_cnt, cnt.size() * 4, hipMemcpyDeviceToHost);
            hipMemcpy(dsum.data(), d_sum, dsum.size() * 4, hipMemcpyDeviceToHost);
            for (uint32_t c = 0; c < nchunks; c++) {
                if (dsum[c] == hsum[c]) continue;
                // what the wave read differs from the host bytes: is the device memory itself wrong now?
                const uint64_t c0 = (uint64_t) c * chunk, len = std::min<uint64_t>(nbytes, c0 + chunk) - c0;
                std::vector<uint8_t> back(len);
                hipMemcpy(back.data(), d_s + c0, len, hipMemcpyDeviceToHost);
                uint64_t diff = 0, first = len, zeros = 0;
                for (uint64_t i = 0; i < len; i++)
                    if (back[i] != s[c0 + i]) {
                        diff++;
                        zeros += back[i] == 0;
                        if (first == len) first = i;
                    }
                printf("  rep %d variant %d chunk %u: the kernel read checksum %08x, host bytes give %08x; "
                       "copied back afterwards %llu of %llu bytes differ (first at %llu, %llu of them zero)\n",
                       rep, variant, c, dsum[c], hsum[c], (unsigned long long) diff, (unsigned long long) len,
                       (unsigned long long) first, (unsigned long long) zeros);
            }
            int bad = 0;
            for (size_t i = 0; i < ex.size(); i++) {
                const bool b = ex[i] != hex[i] || cnt[i] != hcnt[i];
                if (b && bad_entries + bad < 24)
                    printf("  rep %d chunk %zu state %zu: exit %x want %x, count %u want %u\n", rep, i / S, i % S, ex[i], hex[i],
                           cnt[i], hcnt[i]);
                bad += b;
            }
            bad_runs += bad != 0;
            bad_entries += bad;
        }
        printf("variant %d (%s): %d of %d runs differ from the host replay (%d table entries in all)\n", variant,
               variant == 0 ? "early return" : "single exit", bad_runs, reps, bad_entries);
    }
    return 0;
}
