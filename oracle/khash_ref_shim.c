/*
 * khash_ref_shim.c -- exports the reference's OWN khash int64->int64 map as a
 * shared library so the oracle's map (lattice_oracle.c: hpl_i2i_*) and the lattice
 * fixtures can be validated against the real thing.
 *
 * The reference declares its four functions `static inline`
 * (/root/reference/models/khash_int2int.h:8-33), so they cannot be linked
 * directly; this file only forwards to them.  It is compiled against the headers
 * WHERE THEY LIE (-I/root/reference/models, see Makefile); no reference source is
 * copied into this repository.  Output goes to oracle/_ref/ (git-ignored).
 */
#include "khash_int2int.h"

void *khash_ref_init(void) { return khash_int2int_init(); }
void khash_ref_destroy(void *h) { khash_int2int_destroy(h); }
long long khash_ref_get(void *h, long long key, long long dflt) {
    return (long long)khash_int2int_get(h, (khint64_t)key, (khint64_t)dflt);
}
int khash_ref_set(void *h, long long key, long long value) {
    return khash_int2int_set(h, (khint64_t)key, (khint64_t)value);
}
