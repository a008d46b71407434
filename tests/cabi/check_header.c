/* tests/cabi/check_header.c -- TEST INFRASTRUCTURE.  Includes include/pcreg.h from plain C (and, compiled
 * again as C++, from C++), links libpcreg_hip.so and calls the host tier with the header's own prototypes:
 * a drift between a declaration and its definition that symbol names alone would not show (argument
 * order, widths, struct layout) fails here.  Input = the known-answer vector the reference holds in
 * testTransformEstimation.m:2-14 (4 points, ZYX Euler (0.1, 0.2, 0.3), t = (1, 2, 3)).
 * Exit codes: 0 = all checks passed on a GPU; 77 = library loaded, no gfx950 device (every call returned
 * PCREG_E_NODEVICE as documented); anything else = failure. */
#include "pcreg.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

static void eul_zyx(double z, double y, double x, double R[9]) {      /* row-major */
    double cz = cos(z), sz = sin(z), cy = cos(y), sy = sin(y), cx = cos(x), sx = sin(x);
    R[0] = cy * cz; R[1] = sy * sx * cz - sz * cx; R[2] = sy * cx * cz + sz * sx;
    R[3] = cy * sz; R[4] = sy * sx * sz + cz * cx; R[5] = sy * cx * sz - cz * sx;
    R[6] = -sy;     R[7] = cy * sx;                R[8] = cy * cx;
}

int main(void) {
    /* pts of testTransformEstimation.m:2-5, column-major 4 x 3 */
    const double pts[12] = {1, 2, 3, 1,   2, 3, 1, 1,   3, 1, 2, 1};
    double R[9], ptf[12], T[16], d[4];
    const double t[3] = {1, 2, 3};
    int i, c, k, empty = -1, rc, nodev = 0;
    eul_zyx(0.1, 0.2, 0.3, R);
    for (i = 0; i < 4; ++i) for (c = 0; c < 3; ++c) {       /* pts_tf = pts * R + t */
        double s = t[c];
        for (k = 0; k < 3; ++k) s += pts[i + 4 * k] * R[3 * k + c];
        ptf[i + 4 * c] = s;
    }
    printf("pcreg %s\n", pcreg_version());
    rc = pcreg_estimate_transform(ptf, pts, 4, 4, T, &empty);
    if (rc == PCREG_E_NODEVICE) nodev = 1;
    else if (rc != PCREG_OK) { printf("estimate_transform rc %d: %s\n", rc, pcreg_last_error()); return 1; }
    else {
        double err = 0.0;
        if (empty) { printf("estimate_transform returned []\n"); return 1; }
        /* expect T = [R 0; t 1] (column-major 4x4, used as [p 1]*T) */
        for (i = 0; i < 3; ++i) for (c = 0; c < 3; ++c) err = fmax(err, fabs(T[i + 4 * c] - R[3 * i + c]));
        for (c = 0; c < 3; ++c) err = fmax(err, fabs(T[3 + 4 * c] - t[c]));
        err = fmax(err, fabs(T[15] - 1.0));
        printf("estimate_transform max |T - [R 0; t 1]| = %.3g\n", err);
        if (!(err < 1e-12)) return 1;
        rc = pcreg_calc_dists(T, ptf, pts, 4, 4, d);
        if (rc != PCREG_OK) { printf("calc_dists rc %d\n", rc); return 1; }
        for (i = 0; i < 4; ++i) if (!(d[i] < 1e-24)) { printf("calc_dists d[%d] = %g\n", i, d[i]); return 1; }
    }
    {
        pcreg_ransac_opts o;
        int32_t inl[4];
        int ni = -1, ns = -1, mi = -1, failed = -1;
        memset(&o, 0, sizeof o);
        o.minPtNum = 3; o.iterNum = 64; o.thDist = 0.1; o.thInlrRatio = 0.5; o.REFINE = 1; o.VERBOSE = 0; o.seed = 5;
        rc = pcreg_ransac(ptf, pts, 4, 4, &o, NULL, T, inl, &ni, &ns, &mi, &failed, NULL, NULL);
        if (rc == PCREG_E_NODEVICE) nodev = 1;
        else if (rc != PCREG_OK) { printf("ransac rc %d: %s\n", rc, pcreg_last_error()); return 1; }
        else {
            printf("ransac: failed %d, %d inliers, numSuccess %d, maxInliers %d\n", failed, ni, ns, mi);
            if (failed || ni != 4 || mi != 4 || ns != 64) return 1;
            for (i = 0; i < 4; ++i) if (inl[i] != i + 1) return 1;
        }
    }
    if (sizeof(pcreg_ransac_opts) != 40 || sizeof(pcreg_match_opts) != 64) { printf("struct sizes %zu %zu\n", sizeof(pcreg_ransac_opts), sizeof(pcreg_match_opts)); return 1; }
    if (nodev) { printf("no gfx950 device: PCREG_E_NODEVICE (%s)\n", pcreg_last_error()); return 77; }
    printf("ok\n");
    return 0;
}
