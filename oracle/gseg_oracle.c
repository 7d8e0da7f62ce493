/*
 * gseg_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE) for the ground
 * segmentation pre-filter, restating
 * /root/reference/ground_segmentation/src/groundSegmentation.cpp line by line
 * (citations inline).  See slam_oracle.h for its parity status (unpinned).
 */
#define _DEFAULT_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "slam_oracle.h"

void ogseg_default_params(ogseg_params *p)
{ /* groundSegmentation.cpp:31-55 */
    p->rmax = 100.0;
    p->num_seedpoints = 10;
    p->p_l = 10;
    p->p_sf = 1.0;
    p->p_sn = 0.3;
    p->p_tmodel = 5.0;
    p->p_tdata = 5.0;
    p->p_tg = 0.3;
    p->robot_height = 1.2;
    p->max_seed_range = 50;
    p->max_seed_height = 15;
}

typedef struct {
    double range, height;
    int    idx;
} sigpt;

static int cmp_sig(const void *a, const void *b)
{ /* :25-28 compareSignalPoints, ties by bin index */
    const sigpt *x = (const sigpt *)a, *y = (const sigpt *)b;
    if (x->height < y->height) return -1;
    if (x->height > y->height) return 1;
    return x->idx - y->idx;
}

/* :165-185 genGPModel: sig_f and p_l arrive as float */
static double gp_cov(double r1, double r2, float sig_f, float p_l)
{
    float  coeff = (-1 / (2 * p_l * p_l));
    double diff = r1 - r2;
    return (double)sig_f * exp((double)coeff * (diff * diff));
}

/* solves A X = B in place (A m x m, B m x k, row-major) by LU with partial pivoting */
static void lu_solve(double *A, double *B, int m, int k)
{
    for (int c = 0; c < m; c++) {
        int    piv = c;
        double best = fabs(A[c * m + c]);
        for (int r = c + 1; r < m; r++)
            if (fabs(A[r * m + c]) > best) {
                best = fabs(A[r * m + c]);
                piv = r;
            }
        if (piv != c) {
            for (int j = 0; j < m; j++) {
                double t = A[c * m + j];
                A[c * m + j] = A[piv * m + j];
                A[piv * m + j] = t;
            }
            for (int j = 0; j < k; j++) {
                double t = B[c * k + j];
                B[c * k + j] = B[piv * k + j];
                B[piv * k + j] = t;
            }
        }
        for (int r = c + 1; r < m; r++) {
            double f = A[r * m + c] / A[c * m + c];
            if (f == 0.0) continue;
            for (int j = c; j < m; j++) A[r * m + j] -= f * A[c * m + j];
            for (int j = 0; j < k; j++) B[r * k + j] -= f * B[c * k + j];
        }
    }
    for (int c = m - 1; c >= 0; c--)
        for (int j = 0; j < k; j++) {
            double s = B[c * k + j];
            for (int r = c + 1; r < m; r++) s -= A[c * m + r] * B[r * k + j];
            B[c * k + j] = s / A[c * m + c];
        }
}

int ogseg_segment(const ogseg_params *p, const float *xyz, int n, int stride, unsigned char *labels,
                  int *bin_of, unsigned char *sector_model, double *sector_value)
{
    const int NA = OGSEG_NUMBINSA, NL = OGSEG_NUMBINSL;
    int      *count = (int *)calloc((size_t)NA * NL, sizeof(int));
    float    *proto_z = (float *)malloc(sizeof(float) * NA * NL);
    float    *sig_x = (float *)malloc(sizeof(float) * NA * NL), *sig_y = (float *)malloc(sizeof(float) * NA * NL);
    int      *bins = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    unsigned char *state = (unsigned char *)calloc((size_t)NA * NL, 1);
    double   *value = (double *)calloc((size_t)NA * NL, sizeof(double));
    for (int i = 0; i < NA * NL; i++) {
        proto_z[i] = OGSEG_INVALID; /* :83 */
        sig_x[i] = OGSEG_INVALID;   /* :84-85 */
        sig_y[i] = OGSEG_INVALID;
    }
    /* ---- genPolarBinGrid :110-162 */
    const double bsize_rad = (double)((360.0) / NA), bsize_lin = (double)p->rmax / NL;
    for (int i = 0; i < n; i++) {
        const double px = xyz[(size_t)i * stride], py = xyz[(size_t)i * stride + 1], pz = xyz[(size_t)i * stride + 2];
        bins[i] = -1;
        if (sqrt(px * px + py * py + pz * pz) < p->rmax) { /* :126 */
            double ph = (atan2(py, px)) * (180 / M_PI);
            if (ph < 0) ph = 360.0 + ph;
            unsigned bind_rad = (unsigned)floor(ph / bsize_rad);
            if (bind_rad >= (unsigned)NA) bind_rad = NA - 1; /* the reference asserts (:136); clamp */
            const double xyDist = sqrt(px * px + py * py);
            unsigned     bind_lin = (unsigned)floor(xyDist / bsize_lin);
            if (bind_lin >= (unsigned)NL) bind_lin = NL - 1; /* unreachable: 3-D range < rmax */
            const int b = (int)bind_rad * NL + (int)bind_lin;
            bins[i] = b;
            count[b]++;
            if (pz < (double)proto_z[b]) { /* :149 smallest z, first one wins */
                proto_z[b] = (float)pz;
                sig_x[b] = (float)xyDist; /* pcl::PointXY holds floats */
                sig_y[b] = (float)pz;
            }
        }
    }
    int    total_iters = 0;
    sigpt *sig = (sigpt *)malloc(sizeof(sigpt) * NL), *model = (sigpt *)malloc(sizeof(sigpt) * NL);
    double *A = (double *)malloc(sizeof(double) * NL * NL), *Bm = (double *)malloc(sizeof(double) * NL * NL);
    double *fs = (double *)malloc(sizeof(double) * NL), *vf = (double *)malloc(sizeof(double) * NL);
    for (int s = 0; s < NA; s++) { /* segmentGround :187-194 -> sectorINSAC :196 */
        int ns = 0, nm = 0;
        for (int i = 0; i < NL; i++) /* :205-219 */
            if (sig_x[s * NL + i] != (float)OGSEG_INVALID && count[s * NL + i] > 5) {
                sig[ns].range = sig_x[s * NL + i];
                sig[ns].height = sig_y[s * NL + i];
                sig[ns].idx = i;
                ns++;
            }
        qsort(sig, (size_t)ns, sizeof(sigpt), cmp_sig); /* :229 */
        const int npt = ns < p->num_seedpoints ? ns : p->num_seedpoints; /* :235 */
        { /* :242-269: the first npt points of the sorted list that pass the gates become the seed */
            int cur = 0, ctr = 0;
            while (1) {
                if (cur >= ns) break;
                if (sig[cur].range < p->max_seed_range && fabs(sig[cur].height) < p->max_seed_height) {
                    model[nm++] = sig[cur];
                    memmove(&sig[cur], &sig[cur + 1], sizeof(sigpt) * (size_t)(ns - cur - 1));
                    ns--;
                    ctr++;
                } else
                    cur++;
                if (ctr >= npt) break;
            }
        }
        int keep = 1, sufficient = 1;
        if (nm < 2) { /* :272-277 */
            keep = 0;
            sufficient = 0;
        }
        if (ns == 0) keep = 0; /* :289-290 */
        while (keep) { /* :295-377 */
            total_iters++;
            /* f_s = C_XsX (C_XX + sn I)^-1 z ; Vf_s(k,k) = C_XsXs(k,k) - [C_XsX (C_XX+sn I)^-1 C_XXs](k,k) */
            for (int i = 0; i < nm; i++)
                for (int j = 0; j < nm; j++)
                    A[i * nm + j] = gp_cov(model[i].range, model[j].range, (float)p->p_sf, (float)p->p_l) +
                                    (i == j ? p->p_sn : 0.0);
            /* right-hand sides: column 0 = model heights, columns 1..ns = C_XXs (m x ns) */
            const int k = ns + 1;
            for (int i = 0; i < nm; i++) {
                Bm[i * k] = model[i].height;
                for (int j = 0; j < ns; j++)
                    Bm[i * k + 1 + j] = gp_cov(sig[j].range, model[i].range, (float)p->p_sf, (float)p->p_l);
            }
            lu_solve(A, Bm, nm, k);
            for (int j = 0; j < ns; j++) {
                double f = 0, q = 0;
                for (int i = 0; i < nm; i++) {
                    const double c = gp_cov(sig[j].range, model[i].range, (float)p->p_sf, (float)p->p_l);
                    f += c * Bm[i * k];
                    q += c * Bm[i * k + 1 + j];
                }
                fs[j] = f;
                vf[j] = gp_cov(sig[j].range, sig[j].range, (float)p->p_sf, (float)p->p_l) - q;
            }
            const int start_size = nm;
            int       kk = 0; /* :331-369: every candidate is tested against THIS iteration's model */
            while (kk < ns) {
                const double met = (sig[kk].height - fs[kk]) / (sqrt(p->p_sn + vf[kk] * vf[kk]));
                if (vf[kk] < p->p_tmodel && fabs(met) < p->p_tdata) {
                    model[nm++] = sig[kk];
                    memmove(&sig[kk], &sig[kk + 1], sizeof(sigpt) * (size_t)(ns - kk - 1));
                    memmove(&fs[kk], &fs[kk + 1], sizeof(double) * (size_t)(ns - kk - 1));
                    memmove(&vf[kk], &vf[kk + 1], sizeof(double) * (size_t)(ns - kk - 1));
                    ns--;
                } else
                    kk++;
            }
            if (start_size == nm || ns == 0) keep = 0; /* :374-375 */
        }
        for (int i = 0; i < nm; i++) { /* :385-418 bins of the ground model */
            state[s * NL + model[i].idx] = 1;
            value[s * NL + model[i].idx] = model[i].height;
        }
        if (sufficient) /* :428-454 the candidates that stayed out: measured against the GP mean */
            for (int i = 0; i < ns; i++) {
                state[s * NL + sig[i].idx] = 2;
                value[s * NL + sig[i].idx] = fs[i];
            }
    }
    for (int i = 0; i < n; i++) {
        unsigned char lab = OGSEG_DROPPED;
        const int     b = bins[i];
        if (b >= 0 && state[b]) {
            const double z = xyz[(size_t)i * stride + 2];
            if (state[b] == 1) {
                const float h = (float)fabs(value[b] - z); /* :397 */
                if (h < p->p_tg)
                    lab = OGSEG_GROUND;
                else
                    lab = h > p->robot_height ? OGSEG_OVERHEAD : OGSEG_OBSTACLE; /* :406-413 */
            } else {
                const float h = (float)fabs(z - value[b]); /* :437 */
                lab = h > p->robot_height ? OGSEG_OVERHEAD : OGSEG_OBSTACLE;
            }
        }
        labels[i] = lab;
    }
    if (bin_of) memcpy(bin_of, bins, sizeof(int) * (size_t)n);
    if (sector_model) memcpy(sector_model, state, (size_t)NA * NL);
    if (sector_value) memcpy(sector_value, value, sizeof(double) * NA * NL);
    free(count); free(proto_z); free(sig_x); free(sig_y); free(bins); free(state); free(value);
    free(sig); free(model); free(A); free(Bm); free(fs); free(vf);
    return total_iters;
}

/* CCICP::classifyPoints, ccicp2d/src/icpTools.cpp:36-103 with icpTools.h:24-26
 * (NUMBINSGA 1200, RESOLUTION 0.5, GRD_ADJ_THRESH 2): obstacle points are binned on a
 * 1200 x 1200 lattice of 0.5 m cells centred on the sensor; a point is "ground adjacent"
 * (GA) when at least 2 of the 8 cells around its cell are empty.  Points outside the
 * lattice or in its outermost ring of cells are dropped (:60, :72-77). */
void occicp_classify(const float *xyz, int n, int stride, unsigned char *flags)
{
    const int    NB = 1200;
    const double RES = 0.5, offset = (double)NB * RES / 2;
    unsigned char *occ = (unsigned char *)calloc((size_t)NB * NB, 1);
    int           *bin = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        const float cx = xyz[(size_t)i * stride], cy = xyz[(size_t)i * stride + 1];
        /* :57-58  floor((curr_x + offset) / RESOLUTION): float + double -> double */
        const double fx = floor(((double)cx + offset) / RES), fy = floor(((double)cy + offset) / RES);
        bin[i] = -1;
        if (!(fx >= 0 && fx < NB && fy >= 0 && fy < NB)) continue; /* :60 (NaN drops too) */
        bin[i] = (int)fx * NB + (int)fy;
        occ[bin[i]] = 1;
    }
    for (int i = 0; i < n; i++) {
        flags[i] = 255;
        if (bin[i] < 0) continue;
        const int bi = bin[i] / NB, bj = bin[i] % NB;
        if (bi == 0 || bi == NB - 1 || bj == 0 || bj == NB - 1) continue; /* :72-77 */
        int ground = 0;
        for (int q = bi - 1; q <= bi + 1; q++)
            for (int r = bj - 1; r <= bj + 1; r++)
                if (!(q == bi && r == bj) && !occ[q * NB + r]) ground++;
        flags[i] = ground >= 2; /* :96 */
    }
    free(occ);
    free(bin);
}
