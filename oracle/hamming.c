/*
 * oracle/hamming.c — CPU restatement of MapMatcher::match_descriptors
 * (reference src/MapMatcher.cpp:129-163).  TEST INFRASTRUCTURE ONLY.
 * PARITY UNPINNED (no reference fixtures; see rs_oracle.h).
 *
 * Third-party semantics restated (OpenCV 4.x, not in /root/reference):
 *   cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, knn, k=2)
 *   -> cv::batchDistance(..., K=2): for each query row the train rows are
 *   visited in ascending index order; a candidate enters the sorted K-list
 *   only if its distance is strictly smaller than the current K-th entry and
 *   is shifted past entries that are strictly greater.  Hence among equal
 *   distances the LOWER train index ranks first.  Distance = sum of popcounts
 *   of the XOR of the 32 bytes.
 */
#include <stdlib.h>
#include <string.h>

#include "rs_oracle.h"

static int hamming256(const uint8_t* a, const uint8_t* b)
{
    /* four 64-bit words (the sum of byte popcounts, computed word-wise; memcpy: rows are only byte aligned) */
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

int orc_hamming_knn2(const uint8_t* query, int nq, const uint8_t* train, int nt,
                     int32_t* idx0, int32_t* dist0, int32_t* idx1, int32_t* dist1)
{
    if (nq < 0 || nt < 0) return 1;
    if (nt == 0) return 0;
#ifdef ORC_OMP      /* all-cores baseline build only (liboracle_omp.so): queries are independent */
#pragma omp parallel for schedule(static) if (nq > 64)
#endif
    for (int q = 0; q < nq; q++) {
        int bi0 = -1, bd0 = 1 << 30, bi1 = -1, bd1 = 1 << 30;
        const uint8_t* qd = query + (size_t)q * 32;
        for (int t = 0; t < nt; t++) {
            int d = hamming256(qd, train + (size_t)t * 32);
            if (d < bd1) {          /* strict: batchDistance `if (d < distptr[K-1])` */
                if (d < bd0) {      /* shifts only past strictly greater entries */
                    bd1 = bd0; bi1 = bi0;
                    bd0 = d; bi0 = t;
                } else {
                    bd1 = d; bi1 = t;
                }
            }
        }
        idx0[q] = bi0; dist0[q] = bd0;
        idx1[q] = bi1; dist1[q] = (bi1 < 0) ? -1 : bd1;
    }
    return 0;
}

int orc_match_descriptors(const uint8_t* query, int nq, const uint8_t* train, int nt,
                          int max_distance, int32_t* match_query, int32_t* match_train,
                          int32_t* match_count)
{
    *match_count = 0;
    if (nq <= 0 || nt <= 0) return 0;   /* src/MapMatcher.cpp:139-141 */
    const int k = nt >= 2 ? 2 : 1;      /* :145 */
    int count = 0;
#ifdef ORC_OMP
    /* all-cores baseline build: the 2-NN of every query in parallel first, the ordered filter after */
    int32_t* a0 = (int32_t*)malloc(sizeof(int32_t) * 4 * (size_t)nq);
    int32_t *ad0 = a0 + nq, *a1 = a0 + 2 * (size_t)nq, *ad1 = a0 + 3 * (size_t)nq;
    orc_hamming_knn2(query, nq, train, nt, a0, ad0, a1, ad1);
#endif
    for (int q = 0; q < nq; q++) {
        int32_t i0, d0, i1 = 0, d1;
        (void)i1;
#ifdef ORC_OMP
        i0 = a0[q]; d0 = ad0[q]; d1 = ad1[q];
#else
        orc_hamming_knn2(query + (size_t)q * 32, 1, train, nt, &i0, &d0, &i1, &d1);
#endif
        if (d0 > max_distance) continue;                 /* :152 */
        /* :156  d0 > 0.75f*d1 in f32; both are integers <= 256 so 0.75f*d1 is
         * exact and the test equals 4*d0 > 3*d1 */
        if (k > 1 && 4 * d0 > 3 * d1) continue;
        match_query[count] = q;                          /* :159-160 */
        match_train[count] = i0;
        count++;
    }
#ifdef ORC_OMP
    free(a0);
#endif
    *match_count = count;
    return 0;
}
