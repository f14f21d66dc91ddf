// A second extern "C" window onto the REAL reference code: the member functions
// of Serializer<K,V> that the pack side of the shuffle calls,
//   * Serializer::kv_to_bytes      src/serializer.h:348-364
//   * Serializer::get_hash_code    src/serializer.h:464-480 (incl. the concatenation of
//                                  several char* keys, :469-473)
// compiled from the sources where they lie under /root/reference (never copied).
//
// The class's constructor allocates its scratch pointers with mem_aligned_malloc
// (src/memory.cpp, which needs the configure-generated mimir_config.h and cannot
// be built here), so no object is constructed: the methods are called on raw
// storage whose fields are set by hand (`private` opened for this one include),
// with the key scratch buffer the constructor would have allocated
// (MAX_RECORD_SIZE bytes, :301-303) supplied by this file.  The method bodies
// that run are the reference's own.  get_hash_code names LOG_ERROR, hence the
// reference's globals.cpp / config.cpp and the image's MPI library at link time
// (oracle/Makefile); nothing of MPI is ever called.
//
// TEST INFRASTRUCTURE ONLY: used to pin oracle/mimir_oracle.c and to generate
// tests/golden/pack_vectors.json.  Built by oracle/Makefile into oracle/_ref/.
#include <stdint.h>
#include <string.h>

#include "globals.h" // all from -I/root/reference/src
#include "hash.h"
#include "memory.h"
// everything serializer.h includes, first: `private` is opened for its own text only
#include <sstream>
#include <string>
#include <type_traits>
#include <typeinfo>
#include <vector>
#include "log.h"
#include "stat.h"
#include "config.h"
#define private public
#include "serializer.h"
#undef private

using namespace MIMIR_NS;

namespace {

char g_tmpkey[1 << 20]; // what the constructor allocates for keycount > 1 (MAX_RECORD_SIZE)

template <class K, class V>
struct RawSerializer {
    alignas(Serializer<K, V>) char mem[sizeof(Serializer<K, V>)];
    Serializer<K, V> *get(int keycount, int valcount)
    {
        Serializer<K, V> *s = reinterpret_cast<Serializer<K, V> *>(mem);
        s->keycount = keycount;
        s->valcount = valcount;
        s->tmpkey = g_tmpkey;
        s->bufkey = NULL;
        s->bufval = NULL;
        return s;
    }
};

struct Blob16 {
    char b[16];
};

} // namespace

extern "C" {

// <char*, uint64_t>, keycount strings (1 or 2)
int ser_kv_cstr_u64(const char *k0, const char *k1, int keycount, uint64_t val, char *buf, int bufsize)
{
    RawSerializer<char *, uint64_t> raw;
    char *keys[2] = {const_cast<char *>(k0), const_cast<char *>(k1)};
    return raw.get(keycount, 1)->kv_to_bytes(keys, &val, buf, bufsize);
}

uint32_t ser_hash_cstr(const char *k0, const char *k1, int keycount)
{
    RawSerializer<char *, uint64_t> raw;
    char *keys[2] = {const_cast<char *>(k0), const_cast<char *>(k1)};
    return raw.get(keycount, 1)->get_hash_code(keys);
}

// <uint64_t, uint64_t>
int ser_kv_u64_u64(uint64_t key, uint64_t val, char *buf, int bufsize)
{
    RawSerializer<uint64_t, uint64_t> raw;
    return raw.get(1, 1)->kv_to_bytes(&key, &val, buf, bufsize);
}

uint32_t ser_hash_u64(uint64_t key)
{
    RawSerializer<uint64_t, uint64_t> raw;
    return raw.get(1, 1)->get_hash_code(&key);
}

// <char, uint64_t> with keycount characters (octree level passes)
int ser_kv_chars_u64(const char *key, int keycount, uint64_t val, char *buf, int bufsize)
{
    RawSerializer<char, uint64_t> raw;
    return raw.get(keycount, 1)->kv_to_bytes(const_cast<char *>(key), &val, buf, bufsize);
}

uint32_t ser_hash_chars(const char *key, int keycount)
{
    RawSerializer<char, uint64_t> raw;
    return raw.get(keycount, 1)->get_hash_code(const_cast<char *>(key));
}

// <char*, 16-byte value> (join's SingleVal)
int ser_kv_cstr_blob16(const char *key, const void *val16, char *buf, int bufsize)
{
    RawSerializer<char *, Blob16> raw;
    char *k = const_cast<char *>(key);
    Blob16 v;
    memcpy(&v, val16, 16);
    return raw.get(1, 1)->kv_to_bytes(&k, &v, buf, bufsize);
}

// The per-record work of CollectiveShuffler::write (src/collectiveshuffler.h:131-166) for
// <uint64_t, uint64_t> pairs with the reference's own compiled pieces in the reference's
// order: get_hash_code (-> hashlittle, src/hash.cpp) modulo the rank count
// (src/baseshuffler.h:268-309), then kv_to_bytes into the target's partition at its
// send_offset.  The loop around them is this file's; it stops at the first pair that does
// not fit (where the reference starts an exchange) and returns how many pairs it packed.
// For timing the reference's code on the host beside the C restatement.
long ser_write_loop_u64_u64(const uint64_t *pairs, long n, int nranks, char *send_buffer, long buf_size,
                            int64_t *send_offset)
{
    RawSerializer<uint64_t, uint64_t> raw;
    Serializer<uint64_t, uint64_t> *ser = raw.get(1, 1);
    long i = 0;
    for (; i < n; i++) {
        uint64_t key = pairs[2 * i], val = pairs[2 * i + 1];
        const int target = (int) (ser->get_hash_code(&key) % (uint32_t) nranks);
        char *buffer = send_buffer + (int64_t) target * buf_size + send_offset[target];
        const int kvsize = ser->kv_to_bytes(&key, &val, buffer, (int) (buf_size - send_offset[target]));
        if (kvsize == -1) break;
        send_offset[target] += kvsize;
    }
    return i;
}

} // extern "C"
