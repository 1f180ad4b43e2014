/* Minimal stand-in for the legacy Torch7 C API <TH/TH.h> (TEST INFRASTRUCTURE ONLY).
 *
 * The reference's Mask R-CNN ops -- geometric/maskrcnn/nms/src/nms.c and roialign/roi_align/src/crop_and_resize.c -- are
 * plain C against TH tensors, an API current PyTorch no longer ships.  This header declares just the handful of types and
 * accessors those two files use, so that oracle/build_ref.py can compile them UNMODIFIED, from where they lie, into
 * oracle/_ref/libmaskrcnn_ref.so; oracle/maskrcnn_ref.py fills the structs from numpy arrays.  Nothing here is derived
 * from TH's sources: it is the obvious struct-with-a-data-pointer the call sites imply. */
#ifndef SDN_ORACLE_TH_SHIM_H
#define SDN_ORACLE_TH_SHIM_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define SDN_TH_TENSOR(NAME, T) \
    typedef struct NAME {      \
        T* data;               \
        long size[4];          \
        int ndim;              \
        long capacity;         \
    } NAME;

SDN_TH_TENSOR(THFloatTensor, float)
SDN_TH_TENSOR(THLongTensor, long)
SDN_TH_TENSOR(THIntTensor, int)
SDN_TH_TENSOR(THByteTensor, unsigned char)

#define THArgCheck(cond, argn, msg)                               \
    do {                                                          \
        if (!(cond)) {                                            \
            fprintf(stderr, "THArgCheck %d: %s\n", (argn), (msg)); \
            abort();                                              \
        }                                                         \
    } while (0)

/* the reference passes float tensors to the Long variant of this check: accept any tensor */
static inline int THLongTensor_isContiguous(const void* t) { (void)t; return 1; }

static inline long THFloatTensor_size(const THFloatTensor* t, int d) { return t->size[d]; }
static inline float* THFloatTensor_data(THFloatTensor* t) { return t->data; }
static inline long* THLongTensor_data(THLongTensor* t) { return t->data; }
static inline int* THIntTensor_data(THIntTensor* t) { return t->data; }
static inline unsigned char* THByteTensor_data(THByteTensor* t) { return t->data; }

static inline long sdn_th_numel(const long* size, int ndim)
{
    long n = 1;
    for (int i = 0; i < ndim; i++) n *= size[i];
    return n;
}

static inline THByteTensor* THByteTensor_newWithSize1d(long n)
{
    THByteTensor* t = (THByteTensor*)calloc(1, sizeof(THByteTensor));
    t->data = (unsigned char*)malloc(n > 0 ? n : 1);
    t->size[0] = n;
    t->ndim = 1;
    t->capacity = n;
    return t;
}
static inline void THByteTensor_fill(THByteTensor* t, unsigned char v) { memset(t->data, v, sdn_th_numel(t->size, t->ndim)); }
static inline void THByteTensor_free(THByteTensor* t)
{
    free(t->data);
    free(t);
}

/* the caller (oracle/maskrcnn_ref.py) owns the storage; resizing only records the shape and checks the capacity */
static inline void THFloatTensor_resize4d(THFloatTensor* t, long a, long b, long c, long d)
{
    if (a * b * c * d > t->capacity) {
        fprintf(stderr, "THFloatTensor_resize4d: capacity %ld < %ld\n", t->capacity, a * b * c * d);
        abort();
    }
    t->size[0] = a;
    t->size[1] = b;
    t->size[2] = c;
    t->size[3] = d;
    t->ndim = 4;
}
static inline void THFloatTensor_zero(THFloatTensor* t) { memset(t->data, 0, sizeof(float) * sdn_th_numel(t->size, t->ndim)); }

#endif
