/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C, fp32, single thread) of the
 * reference's algorithms on the Faster R-CNN + ARD hot path.  It exists to CHECK the HIP product
 * path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) and is never called, linked
 * or imported by anything under abr_iod_amd/.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py against
 * golden vectors generated from the reference's own code in the build container
 * (tests/golden/make_golden.py; the reference's csrc is compiled unmodified by `make ref`).
 *
 * Each function cites the reference file:line (under /root/reference/maskrcnn_benchmark/) whose
 * arithmetic it follows.  Build with -ffp-contract=off: the reference's g++ -O2 x86-64 build has no
 * FMA contraction and integer tap indices must be reproduced bit-for-bit.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * ROIAlign -- csrc/cpu/ROIAlign_cpu.cpp:17-219 (forward); csrc/cuda/ROIAlign_cuda.cu:125-254 (the
 * backward formula; the reference has NO CPU backward, csrc/ROIAlign.h:44).
 * One bilinear sample = 4 taps (flat indices into one H*W plane) + 4 weights.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int p[4]; float w[4]; int valid; } tap_t;

static tap_t make_tap(int H, int W, float y, float x) {
    tap_t t;
    memset(&t, 0, sizeof t);
    /* ROIAlign_cpu.cpp:48 : samples outside [-1,H] x [-1,W] contribute nothing */
    if (y < -1.0 || y > H || x < -1.0 || x > W) return t;
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;   /* :75-80 */
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;   /* :82-87 */
    float ly = y - yl, lx = x - xl;
    float hy = 1.f - ly, hx = 1.f - lx;
    t.w[0] = hy * hx; t.w[1] = hy * lx; t.w[2] = ly * hx; t.w[3] = ly * lx;  /* :92 */
    t.p[0] = yl * W + xl; t.p[1] = yl * W + xh; t.p[2] = yh * W + xl; t.p[3] = yh * W + xh;
    t.valid = 1;
    return t;
}

typedef struct { float y0, x0, bh, bw; int gh, gw; int b; } roi_geom_t;

static roi_geom_t roi_geom(const float* r, float scale, int ph, int pw, int sr) {
    roi_geom_t g;
    g.b = (int)r[0];
    float sw = r[1] * scale, sh = r[2] * scale, ew = r[3] * scale, eh = r[4] * scale; /* :144-147 no rounding */
    float rw = fmaxf(ew - sw, 1.f), rh = fmaxf(eh - sh, 1.f);                         /* :154-155 */
    g.x0 = sw; g.y0 = sh;
    g.bh = rh / (float)ph; g.bw = rw / (float)pw;
    g.gh = sr > 0 ? sr : (int)ceilf(rh / ph);                                          /* :160-164 */
    g.gw = sr > 0 ? sr : (int)ceilf(rw / pw);
    return g;
}

static inline float sample_y(const roi_geom_t* g, int ph, int iy) {
    /* :39-41 -- keep this exact association: start + ph*bin + ((iy+.5)*bin)/grid */
    return g->y0 + ph * g->bh + (float)(iy + .5f) * g->bh / (float)g->gh;
}
static inline float sample_x(const roi_geom_t* g, int pw, int ix) {
    return g->x0 + pw * g->bw + (float)(ix + .5f) * g->bw / (float)g->gw;
}

/* feat [B,C,H,W], rois [K,5] -> out [K,C,PH,PW]   (NCHW, as maskrcnn_benchmark._C.roi_align_forward) */
void abr_oracle_roi_align_forward(const float* feat, const float* rois, int K, int C, int H, int W,
                                  float scale, int PH, int PW, int sr, float* out) {
    for (int n = 0; n < K; n++) {
        roi_geom_t g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
        int ns = g.gh * g.gw;
        float count = (float)ns;
        tap_t* taps = (tap_t*)malloc(sizeof(tap_t) * (size_t)ns * PH * PW);
        int q = 0;
        for (int ph = 0; ph < PH; ph++)
            for (int pw = 0; pw < PW; pw++)
                for (int iy = 0; iy < g.gh; iy++)
                    for (int ix = 0; ix < g.gw; ix++)
                        taps[q++] = make_tap(H, W, sample_y(&g, ph, iy), sample_x(&g, pw, ix));
        for (int c = 0; c < C; c++) {
            const float* plane = feat + ((size_t)g.b * C + c) * H * W;
            float* o = out + ((size_t)n * C + c) * PH * PW;
            q = 0;
            for (int bin = 0; bin < PH * PW; bin++) {
                float acc = 0.f;
                for (int s = 0; s < ns; s++, q++) {
                    const tap_t* t = &taps[q];
                    acc += t->w[0] * plane[t->p[0]] + t->w[1] * plane[t->p[1]] +
                           t->w[2] * plane[t->p[2]] + t->w[3] * plane[t->p[3]];      /* :199-202 */
                }
                o[bin] = acc / count;                                                 /* :207 */
            }
        }
        free(taps);
    }
}

/* Tap table only: for each (roi, bin, sample) the 4 flat indices (or -1) -- used to prove the HIP
 * kernel's INTEGER indexing bit-exact independently of float accumulation order.
 * idx_out [K, PH*PW, max_s, 4] int32 (unused slots = -2); returns nothing. */
void abr_oracle_roi_align_taps(const float* rois, int K, int H, int W, float scale, int PH, int PW,
                               int sr, int max_s, int32_t* idx_out, int32_t* grid_out) {
    for (int n = 0; n < K; n++) {
        roi_geom_t g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
        grid_out[2 * n] = g.gh; grid_out[2 * n + 1] = g.gw;
        for (int ph = 0; ph < PH; ph++)
            for (int pw = 0; pw < PW; pw++) {
                int32_t* o = idx_out + (((size_t)n * PH + ph) * PW + pw) * max_s * 4;
                for (int s = 0; s < max_s * 4; s++) o[s] = -2;
                int s = 0;
                for (int iy = 0; iy < g.gh; iy++)
                    for (int ix = 0; ix < g.gw; ix++, s++) {
                        if (s >= max_s) continue;
                        tap_t t = make_tap(H, W, sample_y(&g, ph, iy), sample_x(&g, pw, ix));
                        for (int k = 0; k < 4; k++) o[4 * s + k] = t.valid ? t.p[k] : -1;
                    }
            }
    }
}

/* grad [K,C,PH,PW] -> grad_feat [B,C,H,W] (zeroed here).  Formula of ROIAlign_cuda.cu:178-254:
 * every sample adds grad*w_k/count to its 4 taps.  Deterministic order (the CUDA one is atomics). */
void abr_oracle_roi_align_backward(const float* grad, const float* rois, int K, int B, int C, int H,
                                   int W, float scale, int PH, int PW, int sr, float* grad_feat) {
    memset(grad_feat, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < K; n++) {
        roi_geom_t g = roi_geom(rois + 5 * n, scale, PH, PW, sr);
        float count = (float)(g.gh * g.gw);
        for (int ph = 0; ph < PH; ph++)
            for (int pw = 0; pw < PW; pw++)
                for (int iy = 0; iy < g.gh; iy++)
                    for (int ix = 0; ix < g.gw; ix++) {
                        tap_t t = make_tap(H, W, sample_y(&g, ph, iy), sample_x(&g, pw, ix));
                        if (!t.valid) continue;
                        for (int c = 0; c < C; c++) {
                            float gv = grad[(((size_t)n * C + c) * PH + ph) * PW + pw];
                            float* plane = grad_feat + ((size_t)g.b * C + c) * H * W;
                            for (int k = 0; k < 4; k++) plane[t.p[k]] += gv * t.w[k] / count;  /* :239-249 */
                        }
                    }
    }
}

/* ------------------------------------------------------------------------------------------------
 * NMS -- csrc/cpu/nms_cpu.cpp:5-75 (suppress when ovr >= thr, :60); csrc/cuda/nms.cu:60 uses '>'.
 * boxes [n,4] xyxy, scores [n]; keep_out [n] int64 receives ascending ORIGINAL indices of the
 * survivors (nms_cpu.cpp:66 nonzero(suppressed==0)); returns their count.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float s; int64_t i; } sc_t;
static int sc_cmp(const void* a, const void* b) {
    const sc_t *x = (const sc_t*)a, *y = (const sc_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i); /* stable: ties keep input order */
}

int64_t abr_oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int strict_gt,
                       int64_t* keep_out) {
    if (n == 0) return 0;
    sc_t* ord = (sc_t*)malloc(sizeof(sc_t) * n);
    float* area = (float*)malloc(sizeof(float) * n);
    uint8_t* dead = (uint8_t*)calloc(n, 1);
    for (int64_t i = 0; i < n; i++) {
        ord[i].s = scores[i]; ord[i].i = i;
        const float* b = boxes + 4 * i;
        area[i] = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);                               /* :22 */
    }
    qsort(ord, n, sizeof(sc_t), sc_cmp);
    for (int64_t a = 0; a < n; a++) {
        int64_t i = ord[a].i;
        if (dead[i]) continue;
        const float* bi = boxes + 4 * i;
        for (int64_t c = a + 1; c < n; c++) {
            int64_t j = ord[c].i;
            if (dead[j]) continue;
            const float* bj = boxes + 4 * j;
            float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
            float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
            float w = fmaxf(0.f, xx2 - xx1 + 1), h = fmaxf(0.f, yy2 - yy1 + 1);       /* :56-57 */
            float inter = w * h;
            float ovr = inter / (area[i] + area[j] - inter);                           /* :59 */
            if (strict_gt ? (ovr > thr) : (ovr >= thr)) dead[j] = 1;                   /* :60 / nms.cu:60 */
        }
    }
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) if (!dead[i]) keep_out[k++] = i;
    free(ord); free(area); free(dead);
    return k;
}

/* ------------------------------------------------------------------------------------------------
 * boxlist_iou -- structures/boxlist_ops.py:53-88 (TO_REMOVE = 1).  a [G,4], b [n,4] -> iou [G,n]
 * ---------------------------------------------------------------------------------------------- */
void abr_oracle_box_iou(const float* a, int G, const float* b, int n, float* iou) {
    for (int g = 0; g < G; g++) {
        const float* p = a + 4 * g;
        float area1 = (p[2] - p[0] + 1) * (p[3] - p[1] + 1);
        for (int j = 0; j < n; j++) {
            const float* q = b + 4 * j;
            float area2 = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
            float lx = fmaxf(p[0], q[0]), ly = fmaxf(p[1], q[1]);
            float rx = fminf(p[2], q[2]), ry = fminf(p[3], q[3]);
            float w = fmaxf(rx - lx + 1, 0.f), h = fmaxf(ry - ly + 1, 0.f);            /* :81 clamp(min=0) */
            float inter = w * h;
            iou[(size_t)g * n + j] = inter / (area1 + area2 - inter);                  /* :87 */
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Matcher -- modeling/matcher.py:42-112.  iou [G,n] -> matches [n] int64
 * (-1 below low, -2 between thresholds; low-quality: every column that ties a row's max gets its
 * own argmax back, :83-112).  argmax ties resolve to the FIRST row (torch.max on CPU).
 * ---------------------------------------------------------------------------------------------- */
void abr_oracle_matcher(const float* iou, int G, int n, float hi, float lo, int allow_low_quality,
                        int64_t* matches) {
    int64_t* all = (int64_t*)malloc(sizeof(int64_t) * n);
    for (int j = 0; j < n; j++) {
        float best = iou[j]; int64_t bi = 0;
        for (int g = 1; g < G; g++) if (iou[(size_t)g * n + j] > best) { best = iou[(size_t)g * n + j]; bi = g; }
        all[j] = bi;
        matches[j] = best < lo ? -1 : (best < hi ? -2 : bi);                           /* :68-75 */
    }
    if (allow_low_quality) {
        for (int g = 0; g < G; g++) {
            float rowmax = iou[(size_t)g * n];
            for (int j = 1; j < n; j++) rowmax = fmaxf(rowmax, iou[(size_t)g * n + j]);
            for (int j = 0; j < n; j++) if (iou[(size_t)g * n + j] == rowmax) matches[j] = all[j];  /* :93-112 */
        }
    }
    free(all);
}

/* ------------------------------------------------------------------------------------------------
 * BoxCoder -- modeling/box_coder.py:22-95
 * ---------------------------------------------------------------------------------------------- */
void abr_oracle_box_encode(const float* gt, const float* ex, int n, const float* wts, float* out) {
    for (int i = 0; i < n; i++) {
        const float *r = gt + 4 * i, *p = ex + 4 * i;
        float ew = p[2] - p[0] + 1, eh = p[3] - p[1] + 1;                              /* :33-36 */
        float ecx = p[0] + 0.5f * ew, ecy = p[1] + 0.5f * eh;
        float gw = r[2] - r[0] + 1, gh = r[3] - r[1] + 1;
        float gcx = r[0] + 0.5f * gw, gcy = r[1] + 0.5f * gh;
        out[4 * i + 0] = wts[0] * (gcx - ecx) / ew;                                    /* :44-47 */
        out[4 * i + 1] = wts[1] * (gcy - ecy) / eh;
        out[4 * i + 2] = wts[2] * logf(gw / ew);
        out[4 * i + 3] = wts[3] * logf(gh / eh);
    }
}

/* deltas [n, 4k], boxes [n,4] -> out [n,4k] */
void abr_oracle_box_decode(const float* deltas, const float* boxes, int n, int k, const float* wts,
                           float* out) {
    const float clip = (float)log(1000.0 / 16);                                        /* box_coder.py:20 */
    for (int i = 0; i < n; i++) {
        const float* b = boxes + 4 * i;
        float w = b[2] - b[0] + 1, h = b[3] - b[1] + 1;                                /* :66-69 */
        float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * h;
        for (int c = 0; c < k; c++) {
            const float* d = deltas + (size_t)i * 4 * k + 4 * c;
            float dx = d[0] / wts[0], dy = d[1] / wts[1];
            float dw = fminf(d[2] / wts[2], clip), dh = fminf(d[3] / wts[3], clip);    /* :77-78 */
            float pcx = dx * w + cx, pcy = dy * h + cy;
            float pw = expf(dw) * w, phh = expf(dh) * h;
            float* o = out + (size_t)i * 4 * k + 4 * c;
            o[0] = pcx - 0.5f * pw; o[1] = pcy - 0.5f * phh;                           /* :88-94 */
            o[2] = pcx + 0.5f * pw - 1; o[3] = pcy + 0.5f * phh - 1;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Anchors -- modeling/rpn/anchor_generator.py:215-284 (cell anchors, float64 + numpy round =
 * half-to-even, :272-273) and :84-110 (grid + visibility).
 * ---------------------------------------------------------------------------------------------- */
void abr_oracle_cell_anchors(int stride, const double* sizes, int ns, const double* ratios, int nr,
                             float* out /* [nr*ns,4] */) {
    double bw = stride, bh = stride, xc = 0.5 * (bw - 1), yc = 0.5 * (bh - 1);         /* :236,:241-248 */
    double area = bw * bh;
    int q = 0;
    for (int r = 0; r < nr; r++) {
        double ws = nearbyint(sqrt(area / ratios[r]));                                 /* :272 np.round */
        double hs = nearbyint(ws * ratios[r]);                                         /* :273 */
        double x1 = xc - 0.5 * (ws - 1), y1 = yc - 0.5 * (hs - 1);                     /* _mkanchors :251-264 */
        double x2 = xc + 0.5 * (ws - 1), y2 = yc + 0.5 * (hs - 1);
        double w0 = x2 - x1 + 1, h0 = y2 - y1 + 1, cx = x1 + 0.5 * (w0 - 1), cy = y1 + 0.5 * (h0 - 1);
        for (int s = 0; s < ns; s++, q++) {
            double sc = sizes[s] / stride;                                             /* :227 */
            double w = w0 * sc, h = h0 * sc;                                           /* _scale_enum :278-284 */
            out[4 * q + 0] = (float)(cx - 0.5 * (w - 1)); out[4 * q + 1] = (float)(cy - 0.5 * (h - 1));
            out[4 * q + 2] = (float)(cx + 0.5 * (w - 1)); out[4 * q + 3] = (float)(cy + 0.5 * (h - 1));
        }
    }
}

/* grid anchors [H*W*A,4] (location-major, anchor-minor :91-93) + visibility for image (ih, iw) */
void abr_oracle_grid_anchors(const float* cell, int A, int H, int W, int stride, int ih, int iw,
                             int straddle, float* out, uint8_t* vis) {
    size_t q = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int a = 0; a < A; a++, q++) {
                float sx = (float)(x * stride), sy = (float)(y * stride);
                float* o = out + 4 * q;
                o[0] = sx + cell[4 * a]; o[1] = sy + cell[4 * a + 1];
                o[2] = sx + cell[4 * a + 2]; o[3] = sy + cell[4 * a + 3];
                if (straddle >= 0)                                                     /* :100-107 */
                    vis[q] = o[0] >= -straddle && o[1] >= -straddle && o[2] < iw + straddle && o[3] < ih + straddle;
                else
                    vis[q] = 1;
            }
}

/* ------------------------------------------------------------------------------------------------
 * Sigmoid focal loss -- csrc/cuda/SigmoidFocalLoss_cuda.cu:20-101 (the only native implementation;
 * note the reference evaluates the `1.`-literal sub-expressions in double before narrowing).
 * ---------------------------------------------------------------------------------------------- */
void abr_oracle_sigmoid_focal_forward(const float* logits, const int32_t* targets, int N, int C,
                                      float gamma, float alpha, float* losses) {
    for (int i = 0; i < N * C; i++) {
        int n = i / C, d = i % C, t = targets[n];
        float c1 = (t == d + 1), c2 = (t >= 0) & (t != d + 1);
        float zn = (float)(1.0 - alpha), zp = alpha, x = logits[i];
        float p = (float)(1. / (1. + expf(-x)));
        float term1 = (float)(powf((float)(1. - p), gamma) * logf(fmaxf(p, FLT_MIN)));
        float term2 = (float)(powf(p, gamma) *
                              (-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))));
        float l = 0.f;
        l += -c1 * term1 * zp;
        l += -c2 * term2 * zn;
        losses[i] = l;
    }
}

void abr_oracle_sigmoid_focal_backward(const float* logits, const int32_t* targets, const float* d_losses,
                                       int N, int C, float gamma, float alpha, float* d_logits) {
    for (int i = 0; i < N * C; i++) {
        int n = i / C, d = i % C, t = targets[n];
        float c1 = (t == d + 1), c2 = (t >= 0) & (t != d + 1);
        float zn = (float)(1.0 - alpha), zp = alpha, x = logits[i];
        float p = (float)(1. / (1. + expf(-x)));
        float term1 = (float)(powf((float)(1. - p), gamma) * (1. - p - (p * gamma * logf(fmaxf(p, FLT_MIN)))));
        float term2 = (float)(powf(p, gamma) *
                              ((-1. * x * (x >= 0) - logf((float)(1. + expf((float)(x - 2. * x * (x >= 0)))))) *
                                   (1. - p) * gamma - p));
        float g = 0.f;
        g += -c1 * term1 * zp;
        g += -c2 * term2 * zn;
        d_logits[i] = g * d_losses[i];
    }
}

/* ------------------------------------------------------------------------------------------------
 * smooth_l1_loss -- layers/smooth_l1_loss.py:6-17 ; returns the SUM (caller divides for the mean)
 * ---------------------------------------------------------------------------------------------- */
double abr_oracle_smooth_l1_sum(const float* x, const float* t, int64_t n, float beta, float* grad) {
    double s = 0;
    for (int64_t i = 0; i < n; i++) {
        float d = x[i] - t[i], a = fabsf(d);
        s += a < beta ? 0.5f * a * a / beta : a - 0.5f * beta;
        if (grad) grad[i] = a < beta ? d / beta : (d > 0 ? 1.f : (d < 0 ? -1.f : 0.f));
    }
    return s;
}
