/*
 * slam_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See slam_oracle.h for scope and parity status.  Every function cites the
 * reference text (file:line under /root/reference) it restates.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -fopenmp slam_oracle.c -lm
 * (-ffp-contract=off: the reference is built by catkin for x86-64 without
 * FMA, so every a*b+c below rounds twice, as there.)
 */
#include "slam_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ====================================================================== */
/* kd-tree: kdtree.cpp                                                    */
/* ====================================================================== */

#define OKD_BUCKET 12 /* kdtree.h:132 */

typedef struct {
    int   cut_dim;
    float cut_val, cut_left, cut_right;
    int   l, u;
    float lo[2], hi[2]; /* box */
    int   left, right;  /* node ids, -1 = none */
} okd_node;

struct okd_tree {
    int       n;
    const float *src;   /* the_data, caller-owned copy below */
    float    *data;     /* copy of input, [n][2] */
    float    *rdata;    /* rearranged into leaf order */
    int      *ind;
    okd_node *nodes;
    int       n_nodes, cap_nodes;
    int       root;
};

/* kdtree.cpp:235-270 spread_in_coordinate: min/max of one coordinate */
static void okd_spread(const okd_tree *t, int c, int l, int u, float *lo, float *hi)
{
    float smin = t->data[2 * t->ind[l] + c], smax = smin;
    for (int i = l + 1; i <= u; i++) {
        float v = t->data[2 * t->ind[i] + c];
        if (v < smin) smin = v;
        if (v > smax) smax = v;
    }
    *lo = smin;
    *hi = smax;
}

/* kdtree.cpp:295-318 select_on_coordinate_value: the swap order fixes the
 * leaf order and so the visit order of ties, keep it as written there. */
static int okd_select_value(okd_tree *t, int c, float alpha, int l, int u)
{
    int lb = l, ub = u;
    while (lb < ub) {
        if (t->data[2 * t->ind[lb] + c] <= alpha) {
            lb++;
        } else {
            int tmp = t->ind[lb];
            t->ind[lb] = t->ind[ub];
            t->ind[ub] = tmp;
            ub--;
        }
    }
    return (t->data[2 * t->ind[lb] + c] <= alpha) ? lb : lb - 1;
}

static int okd_new_node(okd_tree *t)
{
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes = t->cap_nodes ? 2 * t->cap_nodes : 64;
        t->nodes = (okd_node *)realloc(t->nodes, sizeof(okd_node) * (size_t)t->cap_nodes);
    }
    okd_node *nd = &t->nodes[t->n_nodes];
    memset(nd, 0, sizeof(*nd));
    nd->left = nd->right = -1;
    return t->n_nodes++;
}

/* kdtree.cpp:119-233 build_tree_for_range.  `pbox`/`pcut` carry the parent's
 * box as it stood when the child is created (before the post-build union). */
static int okd_build_range(okd_tree *t, int l, int u, int has_parent, int pcut,
                           const float plo[2], const float phi[2])
{
    if (u < l) return -1;
    int id = okd_new_node(t);
    float lo[2], hi[2];

    if ((u - l) <= OKD_BUCKET) {
        for (int i = 0; i < 2; i++) okd_spread(t, i, l, u, &lo[i], &hi[i]);
        okd_node *nd = &t->nodes[id];
        nd->cut_dim = 0;
        nd->cut_val = 0.0f;
        nd->l = l;
        nd->u = u;
        memcpy(nd->lo, lo, sizeof lo);
        memcpy(nd->hi, hi, sizeof hi);
        return id;
    }

    int   c = -1;
    float maxspread = 0.0f;
    for (int i = 0; i < 2; i++) {
        if (!has_parent || pcut == i) {
            okd_spread(t, i, l, u, &lo[i], &hi[i]);
        } else {
            lo[i] = plo[i];
            hi[i] = phi[i];
        }
        float spread = hi[i] - lo[i];
        if (spread > maxspread) {
            maxspread = spread;
            c = i;
        }
    }
    if (c < 0) {
        /* all points identical: the reference indexes coordinate -1 and then
         * recurses without end; the oracle makes an oversized leaf instead. */
        for (int i = 0; i < 2; i++) okd_spread(t, i, l, u, &lo[i], &hi[i]);
        okd_node *nd = &t->nodes[id];
        nd->l = l;
        nd->u = u;
        memcpy(nd->lo, lo, sizeof lo);
        memcpy(nd->hi, hi, sizeof hi);
        return id;
    }

    float sum = 0.0f; /* kdtree.cpp:178-186: float running sum, then mean */
    for (int k = l; k <= u; k++) sum += t->data[2 * t->ind[k] + c];
    float average = sum / (float)(u - l + 1);
    int   m = okd_select_value(t, c, average, l, u);
    if (m >= u) m = u - 1; /* guards the reference's unbounded recursion */
    if (m < l) m = l;

    t->nodes[id].cut_dim = c;
    t->nodes[id].l = l;
    t->nodes[id].u = u;
    int left = okd_build_range(t, l, m, 1, c, lo, hi);
    int right = okd_build_range(t, m + 1, u, 1, c, lo, hi);
    okd_node *nd = &t->nodes[id]; /* nodes may have moved */
    nd->left = left;
    nd->right = right;
    const okd_node *L = left >= 0 ? &t->nodes[left] : NULL;
    const okd_node *Rn = right >= 0 ? &t->nodes[right] : NULL;
    if (!Rn) {
        memcpy(nd->lo, L->lo, sizeof lo);
        memcpy(nd->hi, L->hi, sizeof hi);
        nd->cut_val = L->hi[c];
        nd->cut_left = nd->cut_right = nd->cut_val;
    } else if (!L) {
        memcpy(nd->lo, Rn->lo, sizeof lo);
        memcpy(nd->hi, Rn->hi, sizeof hi);
        nd->cut_val = Rn->hi[c];
        nd->cut_left = nd->cut_right = nd->cut_val;
    } else {
        nd->cut_right = Rn->lo[c];
        nd->cut_left = L->hi[c];
        nd->cut_val = (float)((nd->cut_left + nd->cut_right) / 2.0);
        for (int i = 0; i < 2; i++) {
            nd->hi[i] = L->hi[i] > Rn->hi[i] ? L->hi[i] : Rn->hi[i];
            nd->lo[i] = L->lo[i] < Rn->lo[i] ? L->lo[i] : Rn->lo[i];
        }
    }
    return id;
}

okd_tree *okd_build(const float *xy, int n)
{
    okd_tree *t = (okd_tree *)calloc(1, sizeof(*t));
    t->n = n;
    t->data = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    t->rdata = (float *)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
    t->ind = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    if (n > 0) memcpy(t->data, xy, sizeof(float) * 2 * (size_t)n);
    for (int i = 0; i < n; i++) t->ind[i] = i;
    float z[2] = {0, 0};
    t->root = okd_build_range(t, 0, n - 1, 0, -1, z, z);
    for (int i = 0; i < n; i++) { /* kdtree.cpp:90-102 rearrange=true */
        t->rdata[2 * i] = t->data[2 * t->ind[i]];
        t->rdata[2 * i + 1] = t->data[2 * t->ind[i] + 1];
    }
    return t;
}

void okd_free(okd_tree *t)
{
    if (!t) return;
    free(t->data);
    free(t->rdata);
    free(t->ind);
    free(t->nodes);
    free(t);
}

typedef struct {
    float q[2];
    float ballsize;
    int   have;
    float dis;
    int   idx;
} okd_sr;

static inline float sq(float x) { return x * x; }

/* kdtree.cpp:559-566 */
static inline float dis_from_bnd(float x, float amin, float amax)
{
    if (x > amax) return x - amax;
    if (x < amin) return amin - x;
    return 0.0f;
}

/* kdtree.cpp:515-557 search, :568-581 box_in_search_range,
 * :583-683 process_terminal_node with nn == 1 */
static void okd_search(const okd_tree *t, int id, okd_sr *sr)
{
    const okd_node *nd = &t->nodes[id];
    if (nd->left < 0 && nd->right < 0) {
        float ballsize = sr->ballsize;
        for (int i = nd->l; i <= nd->u; i++) {
            float dis = 0.0f;
            int   early = 0;
            for (int k = 0; k < 2; k++) {
                dis += sq(t->rdata[2 * i + k] - sr->q[k]);
                if (dis > ballsize) {
                    early = 1;
                    break;
                }
            }
            if (early) continue;
            /* first admitted point is pushed, later ones replace it: either
             * way the single result becomes (dis, ind[i]) and the ball
             * shrinks to dis; dis == ballsize replaces (no strict test). */
            sr->have = 1;
            sr->dis = dis;
            sr->idx = t->ind[i];
            ballsize = dis;
        }
        sr->ballsize = ballsize;
        return;
    }
    int   closer, farther;
    float extra;
    float qval = sr->q[nd->cut_dim];
    if (qval < nd->cut_val) {
        closer = nd->left;
        farther = nd->right;
        extra = nd->cut_right - qval;
    } else {
        closer = nd->right;
        farther = nd->left;
        extra = qval - nd->cut_left;
    }
    if (closer >= 0) okd_search(t, closer, sr);
    if (farther >= 0 && sq(extra) < sr->ballsize) {
        const okd_node *f = &t->nodes[farther];
        float dis2 = 0.0f;
        int   in_range = 1;
        for (int i = 0; i < 2; i++) {
            dis2 += sq(dis_from_bnd(sr->q[i], f->lo[i], f->hi[i]));
            if (dis2 > sr->ballsize) {
                in_range = 0;
                break;
            }
        }
        if (in_range) okd_search(t, farther, sr);
    }
}

void okd_nn1(const okd_tree *t, float qx, float qy, float *dis, int *idx)
{
    okd_sr sr;
    sr.q[0] = qx;
    sr.q[1] = qy;
    sr.ballsize = 1.0e38f; /* kdtree.cpp:325 "infinity" */
    sr.have = 0;
    sr.dis = 1.0e38f;
    sr.idx = -1;
    if (t->root >= 0) okd_search(t, t->root, &sr);
    *dis = sr.dis;
    *idx = sr.idx;
}

void obf_nn1(const float *xy, int n, float qx, float qy, float *dis, int *idx)
{
    float best = 0.0f;
    int   bi = -1;
    for (int i = 0; i < n; i++) {
        float d = 0.0f;
        d += sq(xy[2 * i] - qx);
        d += sq(xy[2 * i + 1] - qy);
        if (bi < 0 || d < best) {
            best = d;
            bi = i;
        }
    }
    *dis = bi < 0 ? 1.0e38f : best;
    *idx = bi;
}

void obf_knn(const float *xy, int n, float qx, float qy, int k, int *idx_out)
{
    /* insertion into a sorted list of k, ordered by (dis, idx) */
    float *bd = (float *)malloc(sizeof(float) * (size_t)k);
    int    cnt = 0;
    for (int i = 0; i < n; i++) {
        float d = 0.0f;
        d += sq(xy[2 * i] - qx);
        d += sq(xy[2 * i + 1] - qy);
        int pos = cnt;
        while (pos > 0 && bd[pos - 1] > d) pos--;
        if (pos >= k) continue;
        int last = cnt < k ? cnt : k - 1;
        for (int j = last; j > pos; j--) {
            bd[j] = bd[j - 1];
            idx_out[j] = idx_out[j - 1];
        }
        bd[pos] = d;
        idx_out[pos] = i;
        if (cnt < k) cnt++;
    }
    for (int j = cnt; j < k; j++) idx_out[j] = -1;
    free(bd);
}

/* ====================================================================== */
/* small solves                                                           */
/* ====================================================================== */

/* icpPointToPoint.cpp:159-162: H = U W V^T (matrix.cpp:582-810), R_ = V U^T.
 * V U^T is the orthogonal polar factor of H^T: a rotation by
 * atan2(H01-H10, H00+H11) when det H >= 0, otherwise the reflection
 * [[c, s],[s,-c]] with angle atan2(H01+H10, H00-H11); the reference applies
 * no determinant fix.  Pinned against oracle/_ref (tests/golden/solve2x2). */
void o_p2p_rotation(const double H[4], double R_[4])
{
    double det = H[0] * H[3] - H[1] * H[2];
    if (det >= 0.0) {
        double a = H[0] + H[3], b = H[1] - H[2];
        double n = sqrt(a * a + b * b);
        double c = 1.0, s = 0.0;
        if (n > 0.0) {
            c = a / n;
            s = b / n;
        }
        R_[0] = c;
        R_[1] = -s;
        R_[2] = s;
        R_[3] = c;
    } else {
        double a = H[0] - H[3], b = H[1] + H[2];
        double n = sqrt(a * a + b * b);
        double c = 1.0, s = 0.0;
        if (n > 0.0) {
            c = a / n;
            s = b / n;
        }
        R_[0] = c;
        R_[1] = s;
        R_[2] = s;
        R_[3] = -c;
    }
}

/* matrix.cpp:420-508 (Gauss-Jordan, full pivoting), m = 3, one rhs column */
int o_solve3(double A[9], double b[3])
{
    int indxc[3], indxr[3], ipiv[3] = {0, 0, 0};
    int irow = 0, icol = 0;
    for (int i = 0; i < 3; i++) {
        double big = 0.0;
        for (int j = 0; j < 3; j++)
            if (ipiv[j] != 1)
                for (int k = 0; k < 3; k++)
                    if (ipiv[k] == 0)
                        if (fabs(A[3 * j + k]) >= big) {
                            big = fabs(A[3 * j + k]);
                            irow = j;
                            icol = k;
                        }
        ++ipiv[icol];
        if (irow != icol) {
            for (int l = 0; l < 3; l++) {
                double tmp = A[3 * irow + l];
                A[3 * irow + l] = A[3 * icol + l];
                A[3 * icol + l] = tmp;
            }
            double tmp = b[irow];
            b[irow] = b[icol];
            b[icol] = tmp;
        }
        indxr[i] = irow;
        indxc[i] = icol;
        if (fabs(A[3 * icol + icol]) < 1e-20) return 0;
        double pivinv = 1.0 / A[3 * icol + icol];
        A[3 * icol + icol] = 1.0;
        for (int l = 0; l < 3; l++) A[3 * icol + l] *= pivinv;
        b[icol] *= pivinv;
        for (int ll = 0; ll < 3; ll++)
            if (ll != icol) {
                double dum = A[3 * ll + icol];
                A[3 * ll + icol] = 0.0;
                for (int l = 0; l < 3; l++) A[3 * ll + l] -= A[3 * icol + l] * dum;
                b[ll] -= b[icol] * dum;
            }
    }
    for (int l = 2; l >= 0; l--)
        if (indxr[l] != indxc[l])
            for (int k = 0; k < 3; k++) {
                double tmp = A[3 * k + indxr[l]];
                A[3 * k + indxr[l]] = A[3 * k + indxc[l]];
                A[3 * k + indxc[l]] = tmp;
            }
    return 1;
}

/* icpPointToPlane.cpp:88-95: svd of [[1,-w],[w,1]] = sqrt(1+w^2) * Rot(atan w),
 * so U V^T is that rotation. */
void o_orthonormal_from_omega(double w, double R_[4])
{
    double n = sqrt(1.0 + w * w);
    double c = 1.0 / n, s = w / n;
    R_[0] = c;
    R_[1] = -s;
    R_[2] = s;
    R_[3] = c;
}

/* ====================================================================== */
/* ICP                                                                    */
/* ====================================================================== */

struct oicp_model {
    int       n_ga, n_nga;
    float    *ga, *nga; /* icp.cpp:51-60: model stored as float */
    okd_tree *tree_ga, *tree_nga;
    float    *all;      /* GA then NGA, for the single-class point-to-line mode */
    okd_tree *tree_all;
    double   *normals;  /* 2 per model point (all), or NULL */
};

oicp_model *oicp_create(const double *m_ga, int n_ga, const double *m_nga, int n_nga)
{
    if (n_ga < 0 || n_nga < 0 || (n_ga + n_nga) < 5) return NULL; /* icp.cpp:38-43 */
    oicp_model *m = (oicp_model *)calloc(1, sizeof(*m));
    m->n_ga = n_ga;
    m->n_nga = n_nga;
    m->ga = (float *)malloc(sizeof(float) * 2 * (size_t)(n_ga ? n_ga : 1));
    m->nga = (float *)malloc(sizeof(float) * 2 * (size_t)(n_nga ? n_nga : 1));
    m->all = (float *)malloc(sizeof(float) * 2 * (size_t)(n_ga + n_nga));
    for (int i = 0; i < 2 * n_ga; i++) m->ga[i] = (float)m_ga[i];
    for (int i = 0; i < 2 * n_nga; i++) m->nga[i] = (float)m_nga[i];
    memcpy(m->all, m->ga, sizeof(float) * 2 * (size_t)n_ga);
    memcpy(m->all + 2 * n_ga, m->nga, sizeof(float) * 2 * (size_t)n_nga);
    m->tree_ga = okd_build(m->ga, n_ga);
    m->tree_nga = okd_build(m->nga, n_nga);
    m->tree_all = okd_build(m->all, n_ga + n_nga);
    return m;
}

void oicp_free(oicp_model *m)
{
    if (!m) return;
    okd_free(m->tree_ga);
    okd_free(m->tree_nga);
    okd_free(m->tree_all);
    free(m->ga);
    free(m->nga);
    free(m->all);
    free(m->normals);
    free(m);
}

/* icpPointToPlane.cpp:279-305 (2-D computeNormal): scatter of the k nearest
 * neighbours; the normal is the direction of least spread.  The reference
 * takes column 1 of U from H.svd(U,W,V), whose singular values come out
 * sorted (matrix.cpp:762): for the symmetric PSD 2x2 scatter that column is
 * the eigenvector of the smaller eigenvalue, which the oracle writes in closed
 * form.  Its sign is the svd's business and is irrelevant to the step (A and b
 * flip together).  Pinned against the compiled matrix.cpp up to that sign
 * (tests/test_oracle_solves.py, tests/golden/solve_golden.npz nm_*). */
void o_normal2(const double *nb_xy, int k, double n_out[2])
{
    double mx = 0, my = 0;
    for (int j = 0; j < k; j++) {
        mx += nb_xy[2 * j];
        my += nb_xy[2 * j + 1];
    }
    mx /= (double)k;
    my /= (double)k;
    double sxx = 0, sxy = 0, syy = 0;
    for (int j = 0; j < k; j++) {
        double dx = nb_xy[2 * j] - mx;
        double dy = nb_xy[2 * j + 1] - my;
        sxx += dx * dx;
        sxy += dx * dy;
        syy += dy * dy;
    }
    /* smaller-eigenvalue eigenvector of [[sxx,sxy],[sxy,syy]]:
     * major axis angle th = 0.5*atan2(2 sxy, sxx - syy); normal = (-sin th, cos th) */
    double th = 0.5 * atan2(2.0 * sxy, sxx - syy);
    n_out[0] = -sin(th);
    n_out[1] = cos(th);
}

void oicp_compute_normals(oicp_model *m, int k)
{
    int n = m->n_ga + m->n_nga;
    if (k > n) k = n;
    free(m->normals);
    m->normals = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    int    *nb = (int *)malloc(sizeof(int) * (size_t)(k > 0 ? k : 1));
    double *P = (double *)malloc(sizeof(double) * 2 * (size_t)(k > 0 ? k : 1));
    for (int i = 0; i < n; i++) {
        obf_knn(m->all, n, m->all[2 * i], m->all[2 * i + 1], k, nb);
        for (int j = 0; j < k; j++) {
            P[2 * j] = (double)m->all[2 * nb[j]];
            P[2 * j + 1] = (double)m->all[2 * nb[j] + 1];
        }
        o_normal2(P, k, m->normals + 2 * i);
    }
    free(P);
    free(nb);
}

const double *oicp_normals(const oicp_model *m) { return m->normals; }

/* icpPointToPoint.cpp:69-70: a double expression stored into a
 * std::vector<float>.  Kept out of line and written through memory so the
 * float rounding happens exactly once, as there (gcc 11 -ftree-vectorize was
 * seen to keep the unrounded double for the later (double)q uses otherwise). */
static __attribute__((noinline)) void transform_query(double r00, double r01, double r10,
                                                      double r11, double t0, double t1,
                                                      const double *P, volatile float *q)
{
    q[0] = (float)(r00 * P[0] + r01 * P[1] + t0);
    q[1] = (float)(r10 * P[0] + r11 * P[1] + t1);
}

static void nn_query(const okd_tree *t, const float *xy, int n, int method, float qx, float qy,
                     float *dis, int *idx)
{
    if (method == OICP_NN_BRUTE)
        obf_nn1(xy, n, qx, qy, dis, idx);
    else
        okd_nn1(t, qx, qy, dis, idx);
}

/* icpPointToPoint.cpp:33-172 */
static double fit_step_p2p(const oicp_model *m, const double *t_ga, int n_tga,
                           const double *t_nga, int n_tnga, double R[4], double t[2],
                           const oicp_params *p, int *n_corr, int *corr_idx)
{
    int     total = n_tga + n_tnga;
    /* the correspondence lists p_m, p_t: one block per THREAD, grown on demand and kept (the reference allocates its
     * Matrix temporaries per step; as the CPU baseline of bench.py this port should not pay malloc's arena locks 30 times
     * per scan on every core of the host) */
    static __thread double *scratch = NULL;
    static __thread size_t  scratch_cap = 0;
    size_t                  need = 4 * (size_t)(total ? total : 1);
    if (need > scratch_cap) {
        free(scratch);
        scratch = (double *)malloc(sizeof(double) * need);
        scratch_cap = scratch ? need : 0;
        if (!scratch) {
            *n_corr = 0;
            return -1.0;
        }
    }
    double *pm = scratch, *pt = scratch + need / 2;
    double  mu_m[2] = {0, 0}, mu_t[2] = {0, 0};
    int     in = 0;
    double  r00 = R[0], r01 = R[1], r10 = R[2], r11 = R[3], t0 = t[0], t1 = t[1];

    if (corr_idx)
        for (int i = 0; i < total; i++) corr_idx[i] = -1;

    for (int cls = 0; cls < 2; cls++) {
        const double  *T = cls == 0 ? t_ga : t_nga;
        int            nT = cls == 0 ? n_tga : n_tnga;
        int            nM = cls == 0 ? m->n_ga : m->n_nga;
        const float   *M = cls == 0 ? m->ga : m->nga;
        const okd_tree *tree = cls == 0 ? m->tree_ga : m->tree_nga;
        if (!(nM > 3)) continue; /* :59, :93 */
        for (int i = 0; i < nT; i++) {
            volatile float qv[2];
            transform_query(r00, r01, r10, r11, t0, t1, &T[2 * i], qv);
            float qx = qv[0], qy = qv[1];
            float dis;
            int   idx;
            nn_query(tree, M, nM, p->nn_method, qx, qy, &dis, &idx);
            if ((double)dis < p->indist) { /* :76 float promoted against double inDist */
                pm[2 * in] = (double)M[2 * idx];
                mu_m[0] += pm[2 * in];
                pm[2 * in + 1] = (double)M[2 * idx + 1];
                mu_m[1] += pm[2 * in + 1];
                pt[2 * in] = (double)qx;
                mu_t[0] += pt[2 * in];
                pt[2 * in + 1] = (double)qy;
                mu_t[1] += pt[2 * in + 1];
                if (corr_idx) corr_idx[(cls == 0 ? 0 : n_tga) + i] = idx;
                in++;
            }
        }
    }
    *n_corr = in;
    if (in == 0) return -1.0; /* :128-131 */
    mu_m[0] = mu_m[0] / (double)in;
    mu_m[1] = mu_m[1] / (double)in;
    mu_t[0] = mu_t[0] / (double)in;
    mu_t[1] = mu_t[1] / (double)in;

    /* :155-159  H = ~q_t * q_m, sums in correspondence order */
    double H[4] = {0, 0, 0, 0};
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            double s = 0.0;
            for (int i = 0; i < in; i++)
                s += (pt[2 * i + a] - mu_t[a]) * (pm[2 * i + b] - mu_m[b]);
            H[2 * a + b] = s;
        }
    double R_[4];
    o_p2p_rotation(H, R_);
    /* :163  t_ = ~mu_m - R_*~mu_t */
    double t_[2];
    t_[0] = mu_m[0] - (R_[0] * mu_t[0] + R_[1] * mu_t[1]);
    t_[1] = mu_m[1] - (R_[2] * mu_t[0] + R_[3] * mu_t[1]);
    /* :166-167  R = R_*R ; t = R_*t + t_ (Matrix::operator* sums k = 0,1 from 0) */
    double Rn[4], tn[2];
    Rn[0] = R_[0] * R[0] + R_[1] * R[2];
    Rn[1] = R_[0] * R[1] + R_[1] * R[3];
    Rn[2] = R_[2] * R[0] + R_[3] * R[2];
    Rn[3] = R_[2] * R[1] + R_[3] * R[3];
    tn[0] = (R_[0] * t[0] + R_[1] * t[1]) + t_[0];
    tn[1] = (R_[2] * t[0] + R_[3] * t[1]) + t_[1];
    memcpy(R, Rn, sizeof Rn);
    memcpy(t, tn, sizeof tn);
    /* :170  max(||R_-I||_F, ||t_||) (matrix.cpp:354-360) */
    double a0 = R_[0] - 1.0, a3 = R_[3] - 1.0;
    double nr = sqrt(a0 * a0 + R_[1] * R_[1] + R_[2] * R_[2] + a3 * a3);
    double nt = sqrt(t_[0] * t_[0] + t_[1] * t_[1]);
    return nr > nt ? nr : nt;
}

/* icpPointToPlane.cpp:37-107 (2-D branch): no classes, no inlier gate; the
 * template is GA then NGA.  Own oracle (that file is not compiled upstream). */
static double fit_step_p2l(const oicp_model *m, const double *t_ga, int n_tga,
                           const double *t_nga, int n_tnga, double R[4], double t[2],
                           const oicp_params *p, int *n_corr, int *corr_idx)
{
    int    total = n_tga + n_tnga, nM = m->n_ga + m->n_nga;
    double r00 = R[0], r01 = R[1], r10 = R[2], r11 = R[3], t0 = t[0], t1 = t[1];
    double AtA[9] = {0}, Atb[3] = {0};
    for (int i = 0; i < total; i++) {
        const double *P = i < n_tga ? &t_ga[2 * i] : &t_nga[2 * (i - n_tga)];
        volatile float qv[2];
        transform_query(r00, r01, r10, r11, t0, t1, P, qv);
        float qx = qv[0], qy = qv[1];
        float dis;
        int   idx;
        nn_query(m->tree_all, m->all, nM, p->nn_method, qx, qy, &dis, &idx);
        if (corr_idx) corr_idx[i] = idx;
        double dx = (double)m->all[2 * idx], dy = (double)m->all[2 * idx + 1];
        double nx = m->normals[2 * idx], ny = m->normals[2 * idx + 1];
        double sx = (double)qx, sy = (double)qy;
        double a[3] = {ny * sx - nx * sy, nx, ny};
        double b = nx * dx + ny * dy - nx * sx - ny * sy;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) AtA[3 * r + c] += a[r] * a[c];
            Atb[r] += a[r] * b;
        }
    }
    *n_corr = total;
    if (!o_solve3(AtA, Atb)) return 0.0; /* :85 falls through to "return 0" at :216 */
    double R_[4], t_[2] = {Atb[1], Atb[2]};
    o_orthonormal_from_omega(Atb[0], R_);
    double Rn[4], tn[2];
    Rn[0] = R_[0] * R[0] + R_[1] * R[2];
    Rn[1] = R_[0] * R[1] + R_[1] * R[3];
    Rn[2] = R_[2] * R[0] + R_[3] * R[2];
    Rn[3] = R_[2] * R[1] + R_[3] * R[3];
    tn[0] = (R_[0] * t[0] + R_[1] * t[1]) + t_[0];
    tn[1] = (R_[2] * t[0] + R_[3] * t[1]) + t_[1];
    memcpy(R, Rn, sizeof Rn);
    memcpy(t, tn, sizeof tn);
    double a0 = R_[0] - 1.0, a3 = R_[3] - 1.0;
    double nr = sqrt(a0 * a0 + R_[1] * R_[1] + R_[2] * R_[2] + a3 * a3);
    double nt = sqrt(t_[0] * t_[0] + t_[1] * t_[1]);
    (void)p;
    return nr > nt ? nr : nt;
}

double oicp_fit_step(const oicp_model *m, const double *t_ga, int n_tga,
                     const double *t_nga, int n_tnga, double R[4], double t[2],
                     const oicp_params *p, int *n_corr, int *corr_idx)
{
    if (p->mode == OICP_MODE_P2L)
        return fit_step_p2l(m, t_ga, n_tga, t_nga, n_tnga, R, t, p, n_corr, corr_idx);
    return fit_step_p2p(m, t_ga, n_tga, t_nga, n_tnga, R, t, p, n_corr, corr_idx);
}

int oicp_fit(const oicp_model *m, const double *t_ga, int n_tga,
             const double *t_nga, int n_tnga, double R[4], double t[2],
             const oicp_params *p, double *trace, int *n_corr_last, double *delta_last)
{
    int    steps = 0, nc = 0;
    double d = 0.0;
    if (n_tga + n_tnga < 5) { /* icp.cpp:100-103 */
        if (n_corr_last) *n_corr_last = 0;
        if (delta_last) *delta_last = 0.0;
        return 0;
    }
    for (int iter = 0; iter < p->max_iter; iter++) { /* icp.cpp:116-122 */
        d = oicp_fit_step(m, t_ga, n_tga, t_nga, n_tnga, R, t, p, &nc, NULL);
        if (trace) {
            double *tr = trace + 8 * steps;
            tr[0] = R[0];
            tr[1] = R[1];
            tr[2] = R[2];
            tr[3] = R[3];
            tr[4] = t[0];
            tr[5] = t[1];
            tr[6] = d;
            tr[7] = (double)nc;
        }
        steps++;
        if (d < p->min_delta) break;
    }
    if (n_corr_last) *n_corr_last = nc;
    if (delta_last) *delta_last = d;
    return steps;
}

void oicp_fit_batch(const oicp_model *m, const double *pts, const int *scan_off,
                    const int *scan_nga, int n_scans, double *R, double *t,
                    const oicp_params *p, int *iters, int *n_corr, double *delta,
                    int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < n_scans; s++) {
        int           o = scan_off[s], n = scan_off[s + 1] - o, ng = scan_nga[s];
        const double *ga = pts + 2 * (size_t)o;
        const double *nga = ga + 2 * (size_t)ng;
        int           nc = 0;
        double        d = 0.0;
        int it = oicp_fit(m, ga, ng, nga, n - ng, R + 4 * (size_t)s, t + 2 * (size_t)s, p, NULL,
                          &nc, &d);
        if (iters) iters[s] = it;
        if (n_corr) n_corr[s] = nc;
        if (delta) delta[s] = d;
    }
}

/* icpPointToPoint.cpp:233-316, including dy = ax - bx (:262) and the unused
 * xy sum; the 3x3 inverse goes through Matrix::inv -> solve (matrix.cpp:393). */
void oicp_edge_weight(const double *pm, const double *pt, int n, double eW[9])
{
    double sx = 0, sy = 0, xpy = 0;
    double MZ[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        double ax = pm[2 * i], ay = pm[2 * i + 1], bx = pt[2 * i], by = pt[2 * i + 1];
        double x = (ax + bx) / 2.0, y = (ay + by) / 2.0;
        double dx = ax - bx, dy = ax - bx;
        (void)by;
        sx += x;
        sy += y;
        xpy += x * x + y * y;
        MZ[0] += dx;
        MZ[1] += dy;
        MZ[2] += -y * dx + x * dy;
    }
    double MM[9] = {(double)n, 0, -sy, 0, (double)n, sx, -sy, sx, xpy};
    /* MMi = inv(MM): solve MM * X = I column by column is what Matrix::inv
     * does with a 3-column right-hand side; D = MMi * MZ */
    double D[3];
    {
        double inv[9];
        for (int c = 0; c < 3; c++) {
            double A[9], e[3] = {0, 0, 0};
            memcpy(A, MM, sizeof A);
            e[c] = 1.0;
            o_solve3(A, e);
            for (int r = 0; r < 3; r++) inv[3 * r + c] = e[r];
        }
        for (int r = 0; r < 3; r++)
            D[r] = inv[3 * r] * MZ[0] + inv[3 * r + 1] * MZ[1] + inv[3 * r + 2] * MZ[2];
    }
    double ss = 0;
    for (int i = 0; i < n; i++) {
        double ax = pm[2 * i], ay = pm[2 * i + 1], bx = pt[2 * i], by = pt[2 * i + 1];
        double x = (ax + bx) / 2.0, y = (ay + by) / 2.0;
        double tx = (ax - bx - D[0] + y * D[2]);
        double ty = (ay - by - D[1] - x * D[2]);
        ss += tx * tx + ty * ty;
    }
    ss = ss / (2 * n - 3);
    double sconst = 1.0 / ss;
    for (int i = 0; i < 9; i++) eW[i] = MM[i] * sconst;
}

/* ====================================================================== */
/* grid: mls.cpp:59-150, mls.h:76-97,154-207                              */
/* ====================================================================== */

int ogrid_cell(const ogrid_params *g, float px, float py, int *cx, int *cy)
{
    int offset_x = g->size_x / 2, offset_y = g->size_y / 2; /* mls.cpp:70-71 */
    /* mls.cpp:77-78: float / double + int, C truncation.  Values an int cannot
     * hold (or NaN) are undefined there; x86 yields INT_MIN, i.e. "skip". */
    double fx = (double)px / g->resolution + (double)offset_x;
    double fy = (double)py / g->resolution + (double)offset_y;
    if (!(fx > -2147483648.0 && fx < 2147483648.0)) return -1;
    if (!(fy > -2147483648.0 && fy < 2147483648.0)) return -1;
    int x = (int)fx, y = (int)fy;
    double rng;
    if (g->rolling) {
        rng = (double)sqrtf(px * px + py * py); /* mls.cpp:82: float expression */
    } else {
        double rx = g->pose_x - (double)px, ry = g->pose_y - (double)py; /* :84-86 */
        rng = sqrt(rx * rx + ry * ry);
    }
    /* mls.cpp:90 -- note y is tested against size_x, as there */
    if (x < 0 || y < 0 || x >= g->size_x || y >= g->size_x || rng > g->max_range) return -1;
    if (y >= g->size_y) return -1; /* size_y < size_x would index out of the plane */
    if (cx) *cx = x;
    if (cy) *cy = y;
    return x + g->size_x * y;
}

long ogrid_add_endpoints(const ogrid_params *g, const float *obs, int n_obs,
                         const float *gnd, int n_gnd, int stride,
                         int32_t *hits, int32_t *misses, int *cell_out)
{
    long n = 0;
    for (int i = 0; i < n_obs; i++) {
        int c = ogrid_cell(g, obs[(size_t)i * stride], obs[(size_t)i * stride + 1], NULL, NULL);
        if (cell_out) cell_out[i] = c;
        if (c < 0) continue;
        hits[c] += 1;
        n++;
    }
    for (int i = 0; i < n_gnd; i++) {
        int c = ogrid_cell(g, gnd[(size_t)i * stride], gnd[(size_t)i * stride + 1], NULL, NULL);
        if (cell_out) cell_out[n_obs + i] = c;
        if (c < 0) continue;
        misses[c] += 1;
        n++;
    }
    return n;
}

long ogrid_raycast(const ogrid_params *g, const float *origin_xy, const float *end_xy,
                   int n, int32_t *hits, int32_t *misses)
{
    long upd = 0;
    int  offset_x = g->size_x / 2, offset_y = g->size_y / 2;
    for (int i = 0; i < n; i++) {
        int x1, y1;
        if (ogrid_cell(g, end_xy[2 * i], end_xy[2 * i + 1], &x1, &y1) < 0) continue;
        double fx = (double)origin_xy[2 * i] / g->resolution + (double)offset_x;
        double fy = (double)origin_xy[2 * i + 1] / g->resolution + (double)offset_y;
        if (!(fx > -2147483648.0 && fx < 2147483648.0)) continue;
        if (!(fy > -2147483648.0 && fy < 2147483648.0)) continue;
        int x0 = (int)fx, y0 = (int)fy;
        if (x0 < 0 || y0 < 0 || x0 >= g->size_x || y0 >= g->size_y) continue;
        int dx = abs(x1 - x0), dy = abs(y1 - y0);
        int sx = x1 > x0 ? 1 : -1, sy = y1 > y0 ? 1 : -1;
        int x = x0, y = y0;
        if (dx >= dy) { /* x-major: y steps when the doubled error passes 2dx */
            int e = dx;
            for (int k = 0; k < dx; k++) {
                misses[x + g->size_x * y] += 1;
                upd++;
                x += sx;
                e += 2 * dy;
                if (e >= 2 * dx) {
                    y += sy;
                    e -= 2 * dx;
                }
            }
        } else {
            int e = dy;
            for (int k = 0; k < dy; k++) {
                misses[x + g->size_x * y] += 1;
                upd++;
                y += sy;
                e += 2 * dx;
                if (e >= 2 * dy) {
                    x += sx;
                    e -= 2 * dy;
                }
            }
        }
        hits[x1 + g->size_x * y1] += 1; /* (x,y) == (x1,y1) here */
        upd++;
    }
    return upd;
}

/* The same traversal with OpenMP over beams and atomic increments: the timed
 * CPU baseline of bench.py (counts are sums, so the result equals ogrid_raycast). */
long ogrid_raycast_mt(const ogrid_params *g, const float *origin_xy, const float *end_xy,
                      int n, int32_t *hits, int32_t *misses, int n_threads)
{
    long upd = 0;
    int  offset_x = g->size_x / 2, offset_y = g->size_y / 2;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(static, 256) reduction(+ : upd)
    for (int i = 0; i < n; i++) {
        int x1, y1;
        if (ogrid_cell(g, end_xy[2 * i], end_xy[2 * i + 1], &x1, &y1) < 0) continue;
        double fx = (double)origin_xy[2 * i] / g->resolution + (double)offset_x;
        double fy = (double)origin_xy[2 * i + 1] / g->resolution + (double)offset_y;
        if (!(fx > -2147483648.0 && fx < 2147483648.0)) continue;
        if (!(fy > -2147483648.0 && fy < 2147483648.0)) continue;
        int x0 = (int)fx, y0 = (int)fy;
        if (x0 < 0 || y0 < 0 || x0 >= g->size_x || y0 >= g->size_y) continue;
        int dx = abs(x1 - x0), dy = abs(y1 - y0);
        int sx = x1 > x0 ? 1 : -1, sy = y1 > y0 ? 1 : -1;
        int x = x0, y = y0;
        int du = dx >= dy ? dx : dy, dv = dx >= dy ? dy : dx, e = du;
        for (int k = 0; k < du; k++) {
            __atomic_fetch_add(&misses[x + g->size_x * y], 1, __ATOMIC_RELAXED);
            if (dx >= dy) x += sx; else y += sy;
            e += 2 * dv;
            if (e >= 2 * du) {
                if (dx >= dy) y += sy; else x += sx;
                e -= 2 * du;
            }
        }
        __atomic_fetch_add(&hits[x1 + g->size_x * y1], 1, __ATOMIC_RELAXED);
        upd += du + 1;
    }
    return upd;
}

void o_transform_points(const double *pts, int n, const double R[4], const double t[2],
                        float *out_xy)
{
    for (int i = 0; i < n; i++) {
        volatile float qv[2];
        transform_query(R[0], R[1], R[2], R[3], t[0], t[1], &pts[2 * i], qv);
        out_xy[2 * i] = qv[0];
        out_xy[2 * i + 1] = qv[1];
    }
}

void ogrid_finalize(const ogrid_params *g, const int32_t *hits, const int32_t *misses,
                    double *num_pts, int8_t *occ)
{
    size_t cells = (size_t)g->size_x * (size_t)g->size_y;
    double minp = (double)g->min_cluster_points;
    for (size_t c = 0; c < cells; c++) {
        int32_t h = hits[c], m = misses[c];
        double  v = num_pts[c];
        v = v + g->occupancy_increment * (double)h;
        if (h > 0 && v > minp) occ[c] = 100; /* mls.cpp:101-105 */
        v = v - g->occupancy_decrement * (double)m;
        if (m > 0 && v < minp) occ[c] = 0; /* mls.cpp:137-141 */
        num_pts[c] = v;
    }
}

void ogrid_add_scan_inorder(const ogrid_params *g, const float *obs, int n_obs,
                            const float *gnd, int n_gnd, int stride,
                            double *num_pts, int8_t *drivable, int8_t *occ)
{
    double minp = (double)g->min_cluster_points;
    for (int i = 0; i < n_obs; i++) {
        int c = ogrid_cell(g, obs[(size_t)i * stride], obs[(size_t)i * stride + 1], NULL, NULL);
        if (c < 0) continue;
        num_pts[c] += g->occupancy_increment;
        if (num_pts[c] > minp) {
            drivable[c] = 0;
            occ[c] = 100;
        }
    }
    for (int i = 0; i < n_gnd; i++) {
        int c = ogrid_cell(g, gnd[(size_t)i * stride], gnd[(size_t)i * stride + 1], NULL, NULL);
        if (c < 0) continue;
        num_pts[c] -= g->occupancy_decrement;
        if (num_pts[c] < minp) {
            drivable[c] = 1;
            occ[c] = 0;
        }
    }
}
