// Thin extern "C" window onto the REAL reference code, compiled from the
// sources where they lie under /root/reference (never copied):
//   * hashlittle()                     src/hash.cpp:152
//   * bytestream<T>::to_bytes / size   src/serializer.h:40-223
// Only the header-only, allocation-free static members of serializer.h are
// instantiated, so nothing from the reference's memory.cpp / mimir.cpp (which
// need the configure-generated mimir_config.h) is required.
//
// TEST INFRASTRUCTURE ONLY: used to pin oracle/mimir_oracle.c and to generate
// tests/golden/*.json.  Built by oracle/Makefile into oracle/_ref/.
#include <stdint.h>
#include <string.h>

#include "globals.h" // all four from -I/root/reference/src
#include "hash.h"
#include "memory.h"
#include "serializer.h"

using namespace MIMIR_NS;

extern "C" {

uint32_t ref_hashlittle(const void *key, size_t length, uint32_t initval)
{
    return hashlittle(key, length, initval);
}

// <char*, uint64_t>  (examples/wordcount.cpp:48)
int ref_kv_cstr_u64(const char *key, uint64_t val, char *buf, int bufsize)
{
    char *k = const_cast<char *>(key);
    int kb = bytestream<char *>::to_bytes(&k, 1, buf, bufsize);
    if (kb == -1) return -1;
    int vb = bytestream<uint64_t>::to_bytes(&val, 1, buf + kb, bufsize - kb);
    if (vb == -1) return -1;
    return kb + vb;
}

// <uint64_t, uint64_t>
int ref_kv_u64_u64(uint64_t key, uint64_t val, char *buf, int bufsize)
{
    int kb = bytestream<uint64_t>::to_bytes(&key, 1, buf, bufsize);
    if (kb == -1) return -1;
    int vb = bytestream<uint64_t>::to_bytes(&val, 1, buf + kb, bufsize - kb);
    if (vb == -1) return -1;
    return kb + vb;
}

// <char, uint64_t> with keycount = level  (examples/octree_clustering.cpp:98-106)
int ref_kv_chars_u64(const char *key, int keycount, uint64_t val, char *buf,
                     int bufsize)
{
    int kb = bytestream<char>::to_bytes(const_cast<char *>(key), keycount, buf,
                                        bufsize);
    if (kb == -1) return -1;
    int vb = bytestream<uint64_t>::to_bytes(&val, 1, buf + kb, bufsize - kb);
    if (vb == -1) return -1;
    return kb + vb;
}

// <char*, 16-byte struct>  (examples/join.cpp:21-43: SingleVal)
int ref_kv_cstr_blob16(const char *key, const void *val16, char *buf,
                       int bufsize)
{
    struct Blob16 { char b[16]; };
    char *k = const_cast<char *>(key);
    int kb = bytestream<char *>::to_bytes(&k, 1, buf, bufsize);
    if (kb == -1) return -1;
    Blob16 v;
    memcpy(&v, val16, 16);
    int vb = bytestream<Blob16>::to_bytes(&v, 1, buf + kb, bufsize - kb);
    if (vb == -1) return -1;
    return kb + vb;
}

int ref_key_size_cstr(const char *key)
{
    char *k = const_cast<char *>(key);
    return bytestream<char *>::size(&k, 1);
}

} // extern "C"
