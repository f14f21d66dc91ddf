// Thin extern "C" window onto the REAL reference code, compiled from the
// sources where they lie under /root/reference (never copied):
//   * hashlittle()                     src/hash.cpp:152
//   * bytestream<T>::to_bytes / size   src/serializer.h:40-223
//   * bytestream<T>::from_bytes, Serializer::kv_from_bytes
//                                      src/serializer.h:54-61, :100-129, :376-416
//   * FileParser::to_line              src/fileparser.h:21-41
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
#include "fileparser.h"

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

// The text of one last block through the REAL FileParser::to_line and the
// reader's own stepping rule (src/filereader.h:152-171: advance by the char*
// key's size = strlen + 1, window used up when the step reaches its end), each
// line split with strtok_r and serialised with the REAL bytestream templates as
// examples/wordcount.cpp:66-82 does.  `text` must have one spare byte after
// nbytes (to_line terminates an unterminated last line there) and is modified.
long ref_wordcount_map(char *text, long nbytes, uint64_t value, char *out, long out_cap,
                       uint64_t *rec_off_out, long max_records)
{
    FileParser parser;
    long start = 0, win = nbytes, nrec = 0, opos = 0;
    while (win > 0 && parser.to_line(text + start, (int) win, true) != -1) {
        char *line = text + start;
        const long move = (long) bytestream<char *>::size(&line, 1);
        char *saveptr = NULL;
        char *word = strtok_r(line, " ", &saveptr);
        while (word != NULL) {
            if (strlen(word) < 1024) {
                if (nrec >= max_records) return -1;
                int kb = bytestream<char *>::to_bytes(&word, 1, out + opos, (int) (out_cap - opos));
                if (kb == -1) return -1;
                int vb = bytestream<uint64_t>::to_bytes(&value, 1, out + opos + kb, (int) (out_cap - opos - kb));
                if (vb == -1) return -1;
                rec_off_out[nrec++] = (uint64_t) opos;
                opos += kb + vb;
            }
            word = strtok_r(NULL, " ", &saveptr);
        }
        if (move >= win) {
            win = 0;
        }
        else {
            start += move;
            win -= move;
        }
    }
    rec_off_out[nrec] = (uint64_t) opos;
    return nrec;
}

// ---- receive side: what save_data does with a source's bytes ------------------
// (src/collectiveshuffler.h:563-585): kv_from_bytes record after record until the
// source's count is used up.  Each function walks `nbytes` of wire-form records
// with the REAL parse routines and writes the offset of every record, then the
// end offset; returns the number of records, -1 if max_records is too small, -2 if
// the walk runs past nbytes.  For value-typed K and V the real Serializer object
// is used (its constructor allocates nothing for them, src/serializer.h:300-319);
// for char* keys the object would need the reference's memory.cpp, so the two
// static from_bytes calls are composed exactly as kv_from_bytes does (:398-410).
long ref_parse_u64_u64(char *buf, long nbytes, uint64_t *rec_off, long max_records)
{
    Serializer<uint64_t, uint64_t> ser(1, 1);
    long pos = 0, n = 0;
    while (pos < nbytes) {
        if (n >= max_records) return -1;
        uint64_t *key = NULL, *val = NULL;
        rec_off[n++] = (uint64_t) pos;
        pos += ser.kv_from_bytes(&key, &val, buf + pos, (int) (nbytes - pos));
    }
    rec_off[n] = (uint64_t) pos;
    return pos == nbytes ? n : -2;
}

long ref_parse_chars_u64(char *buf, long nbytes, int keycount, uint64_t *rec_off, long max_records)
{
    Serializer<char, uint64_t> ser(keycount, 1);
    long pos = 0, n = 0;
    while (pos < nbytes) {
        if (n >= max_records) return -1;
        char *key = NULL;
        uint64_t *val = NULL;
        rec_off[n++] = (uint64_t) pos;
        pos += ser.kv_from_bytes(&key, &val, buf + pos, (int) (nbytes - pos));
    }
    rec_off[n] = (uint64_t) pos;
    return pos == nbytes ? n : -2;
}

// <char*, V> with V a value type of vbytes bytes (8: wordcount; 16: join's SingleVal;
// 0: void): key by bytestream<char*>::from_bytes, value by bytestream<V>::from_bytes.
long ref_parse_cstr_fixed(char *buf, long nbytes, int vbytes, uint64_t *rec_off, long max_records)
{
    struct Blob16 { char b[16]; };
    long pos = 0, n = 0;
    while (pos < nbytes) {
        if (n >= max_records) return -1;
        rec_off[n++] = (uint64_t) pos;
        char *key = NULL;
        pos += bytestream<char *>::from_bytes(&key, 1, buf + pos);
        if (vbytes == 8) {
            uint64_t v;
            pos += bytestream<uint64_t>::from_bytes(&v, 1, buf + pos);
        }
        else if (vbytes == 16) {
            Blob16 v;
            pos += bytestream<Blob16>::from_bytes(&v, 1, buf + pos);
        }
        else if (vbytes == 0) {
            pos += bytestream<void>::from_bytes(NULL, 1, buf + pos);
        }
        else {
            return -3;
        }
        if (pos > nbytes) return -2;
    }
    rec_off[n] = (uint64_t) pos;
    return n;
}

} // extern "C"
