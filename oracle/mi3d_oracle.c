/*
 * mi3d_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement (double accumulation, naive loops) of every operator on the
 * 3D U-Net training hot path of fransiskusbudi/multimodal_segmentation_project.  The
 * reference delegates these ops to PyTorch; each function below states the published
 * semantics of the torch op at the reference call site it cites (paths relative to
 * /root/reference).  Pinned by tests/test_oracle_cpu.py against tests/golden (npz files), which
 * tools/gen_golden.py produced by executing the reference itself.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (multimodal_segmentation_project_amd) never does.
 *
 * Layout everywhere: torch's NCDHW, float32 tensors, int64 labels.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX5(n, c, d, h, w, C, D, H, W) (((((size_t)(n) * (C) + (c)) * (D) + (d)) * (H) + (h)) * (W) + (w))

/* nn.Conv3d(Cin,Cout,k,padding=(k-1)/2): models/unet.py:11,15 (k=3) and :62 (k=1).
 * weight (Cout,Cin,k,k,k), cross-correlation, zero padding. */
void orc_conv3d_fwd(const float *x, const float *w, const float *b, float *y,
                    int N, int Cin, int Cout, int D, int H, int W, int k) {
    int p = (k - 1) / 2;
    for (int n = 0; n < N; n++)
        for (int co = 0; co < Cout; co++)
            for (int d = 0; d < D; d++)
                for (int h = 0; h < H; h++)
                    for (int ww = 0; ww < W; ww++) {
                        double acc = b ? b[co] : 0.0;
                        for (int ci = 0; ci < Cin; ci++)
                            for (int i = 0; i < k; i++) {
                                int dd = d + i - p;
                                if (dd < 0 || dd >= D) continue;
                                for (int j = 0; j < k; j++) {
                                    int hh = h + j - p;
                                    if (hh < 0 || hh >= H) continue;
                                    for (int l = 0; l < k; l++) {
                                        int wx = ww + l - p;
                                        if (wx < 0 || wx >= W) continue;
                                        acc += (double)x[IDX5(n, ci, dd, hh, wx, Cin, D, H, W)] *
                                               (double)w[((((size_t)co * Cin + ci) * k + i) * k + j) * k + l];
                                    }
                                }
                            }
                        y[IDX5(n, co, d, h, ww, Cout, D, H, W)] = (float)acc;
                    }
}

/* Backward of the above: gx = conv(gy, flip(w)^T); gw[co,ci,ijk] = sum gy * x_shift; gb = sum gy. */
void orc_conv3d_bwd(const float *x, const float *w, const float *gy, float *gx, float *gw, float *gb,
                    int N, int Cin, int Cout, int D, int H, int W, int k) {
    int p = (k - 1) / 2;
    size_t nx = (size_t)N * Cin * D * H * W, nw = (size_t)Cout * Cin * k * k * k;
    double *dgx = (double *)calloc(nx, sizeof(double));
    double *dgw = (double *)calloc(nw, sizeof(double));
    for (int co = 0; co < Cout; co++) {
        double sb = 0.0;
        for (int n = 0; n < N; n++)
            for (int d = 0; d < D; d++)
                for (int h = 0; h < H; h++)
                    for (int ww = 0; ww < W; ww++) {
                        double g = gy[IDX5(n, co, d, h, ww, Cout, D, H, W)];
                        sb += g;
                        for (int ci = 0; ci < Cin; ci++)
                            for (int i = 0; i < k; i++) {
                                int dd = d + i - p;
                                if (dd < 0 || dd >= D) continue;
                                for (int j = 0; j < k; j++) {
                                    int hh = h + j - p;
                                    if (hh < 0 || hh >= H) continue;
                                    for (int l = 0; l < k; l++) {
                                        int wx = ww + l - p;
                                        if (wx < 0 || wx >= W) continue;
                                        size_t xi = IDX5(n, ci, dd, hh, wx, Cin, D, H, W);
                                        size_t wi = ((((size_t)co * Cin + ci) * k + i) * k + j) * k + l;
                                        dgx[xi] += g * (double)w[wi];
                                        dgw[wi] += g * (double)x[xi];
                                    }
                                }
                            }
                    }
        if (gb) gb[co] = (float)sb;
    }
    if (gx) for (size_t i = 0; i < nx; i++) gx[i] = (float)dgx[i];
    if (gw) for (size_t i = 0; i < nw; i++) gw[i] = (float)dgw[i];
    free(dgx);
    free(dgw);
}

/* nn.BatchNorm3d train mode: models/unet.py:12,16.  Per-channel mean / biased var over N*V;
 * running <- (1-m)*running + m*(mean, unbiased var); eps inside the sqrt. */
void orc_bn_train_fwd(const float *x, const float *gamma, const float *beta, float *rmean, float *rvar,
                      double momentum, double eps, float *y, float *save_mean, float *save_invstd,
                      int N, int C, int64_t V) {
    double M = (double)N * (double)V;
    for (int c = 0; c < C; c++) {
        double s = 0.0;
        for (int n = 0; n < N; n++) {
            const float *p = x + ((size_t)n * C + c) * V;
            for (int64_t v = 0; v < V; v++) s += p[v];
        }
        double mean = s / M, q = 0.0;
        for (int n = 0; n < N; n++) {
            const float *p = x + ((size_t)n * C + c) * V;
            for (int64_t v = 0; v < V; v++) { double d = p[v] - mean; q += d * d; }
        }
        double var = q / M, inv = 1.0 / sqrt(var + eps);
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)inv;
        if (rmean) rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * mean);
        if (rvar) rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * (M > 1 ? q / (M - 1.0) : var));
        for (int n = 0; n < N; n++) {
            const float *p = x + ((size_t)n * C + c) * V;
            float *o = y + ((size_t)n * C + c) * V;
            for (int64_t v = 0; v < V; v++) o[v] = (float)((p[v] - mean) * inv * gamma[c] + beta[c]);
        }
    }
}

/* eval mode: y = (x - running_mean) / sqrt(running_var + eps) * gamma + beta */
void orc_bn_eval_fwd(const float *x, const float *gamma, const float *beta, const float *rmean,
                     const float *rvar, double eps, float *y, int N, int C, int64_t V) {
    for (int n = 0; n < N; n++)
        for (int c = 0; c < C; c++) {
            double inv = 1.0 / sqrt((double)rvar[c] + eps);
            const float *p = x + ((size_t)n * C + c) * V;
            float *o = y + ((size_t)n * C + c) * V;
            for (int64_t v = 0; v < V; v++) o[v] = (float)((p[v] - rmean[c]) * inv * gamma[c] + beta[c]);
        }
}

/* dgamma = sum gy*xhat, dbeta = sum gy, gx = gamma*inv*(gy - dbeta/M - xhat*dgamma/M) */
void orc_bn_train_bwd(const float *x, const float *gy, const float *gamma, const float *save_mean,
                      const float *save_invstd, float *gx, float *ggamma, float *gbeta,
                      int N, int C, int64_t V) {
    double M = (double)N * (double)V;
    for (int c = 0; c < C; c++) {
        double mean = save_mean[c], inv = save_invstd[c], sg = 0.0, sgx = 0.0;
        for (int n = 0; n < N; n++) {
            const float *p = x + ((size_t)n * C + c) * V, *g = gy + ((size_t)n * C + c) * V;
            for (int64_t v = 0; v < V; v++) { sg += g[v]; sgx += g[v] * (p[v] - mean) * inv; }
        }
        if (ggamma) ggamma[c] = (float)sgx;
        if (gbeta) gbeta[c] = (float)sg;
        for (int n = 0; n < N; n++) {
            const float *p = x + ((size_t)n * C + c) * V, *g = gy + ((size_t)n * C + c) * V;
            float *o = gx + ((size_t)n * C + c) * V;
            for (int64_t v = 0; v < V; v++) {
                double xh = (p[v] - mean) * inv;
                o[v] = (float)(gamma[c] * inv * (g[v] - sg / M - xh * sgx / M));
            }
        }
    }
}

/* nn.ReLU(inplace) then nn.Dropout3d(p): models/unet.py:13-14,17-18.  scale[n*C+c] is 0 or 1/(1-p)
 * (all ones in eval / p=0).  Backward masks with the ReLU OUTPUT > 0. */
void orc_relu_drop_fwd(const float *x, const float *scale, float *y, int N, int C, int64_t V) {
    for (size_t nc = 0; nc < (size_t)N * C; nc++)
        for (int64_t v = 0; v < V; v++) {
            float r = x[nc * V + v] > 0.f ? x[nc * V + v] : 0.f;
            y[nc * V + v] = r * (scale ? scale[nc] : 1.f);
        }
}
void orc_relu_drop_bwd(const float *x, const float *scale, const float *gy, float *gx, int N, int C, int64_t V) {
    for (size_t nc = 0; nc < (size_t)N * C; nc++)
        for (int64_t v = 0; v < V; v++)
            gx[nc * V + v] = x[nc * V + v] > 0.f ? gy[nc * V + v] * (scale ? scale[nc] : 1.f) : 0.f;
}

/* nn.MaxPool3d(2,2): models/unet.py:40,71.  Floor output size; backward to the first max in d,h,w scan order. */
void orc_maxpool2_fwd(const float *x, float *y, int N, int C, int D, int H, int W) {
    int Do = D / 2, Ho = H / 2, Wo = W / 2;
    for (size_t nc = 0; nc < (size_t)N * C; nc++)
        for (int d = 0; d < Do; d++)
            for (int h = 0; h < Ho; h++)
                for (int w = 0; w < Wo; w++) {
                    float m = -INFINITY;
                    for (int i = 0; i < 2; i++)
                        for (int j = 0; j < 2; j++)
                            for (int l = 0; l < 2; l++) {
                                float v = x[((nc * D + 2 * d + i) * H + 2 * h + j) * W + 2 * w + l];
                                if (v > m) m = v;
                            }
                    y[((nc * Do + d) * Ho + h) * Wo + w] = m;
                }
}
void orc_maxpool2_bwd(const float *x, const float *gy, float *gx, int N, int C, int D, int H, int W) {
    int Do = D / 2, Ho = H / 2, Wo = W / 2;
    memset(gx, 0, sizeof(float) * (size_t)N * C * D * H * W);
    for (size_t nc = 0; nc < (size_t)N * C; nc++)
        for (int d = 0; d < Do; d++)
            for (int h = 0; h < Ho; h++)
                for (int w = 0; w < Wo; w++) {
                    float m = -INFINITY;
                    size_t arg = 0;
                    for (int i = 0; i < 2; i++)
                        for (int j = 0; j < 2; j++)
                            for (int l = 0; l < 2; l++) {
                                size_t xi = ((nc * D + 2 * d + i) * H + 2 * h + j) * W + 2 * w + l;
                                if (x[xi] > m) { m = x[xi]; arg = xi; }
                            }
                    gx[arg] += gy[((nc * Do + d) * Ho + h) * Wo + w];
                }
}

/* nn.ConvTranspose3d(Cin,Cout,2,stride=2): models/unet.py:56-58,79.  weight (Cin,Cout,2,2,2).
 * out[n,co,2d+i,2h+j,2w+l] = b[co] + sum_ci x[n,ci,d,h,w] * W[ci,co,i,j,l] */
void orc_convT2_fwd(const float *x, const float *w, const float *b, float *y,
                    int N, int Cin, int Cout, int D, int H, int W) {
    int D2 = 2 * D, H2 = 2 * H, W2 = 2 * W;
    for (int n = 0; n < N; n++)
        for (int co = 0; co < Cout; co++)
            for (int d = 0; d < D2; d++)
                for (int h = 0; h < H2; h++)
                    for (int ww = 0; ww < W2; ww++) {
                        double acc = b ? b[co] : 0.0;
                        int i = d & 1, j = h & 1, l = ww & 1;
                        for (int ci = 0; ci < Cin; ci++)
                            acc += (double)x[IDX5(n, ci, d / 2, h / 2, ww / 2, Cin, D, H, W)] *
                                   (double)w[((((size_t)ci * Cout + co) * 2 + i) * 2 + j) * 2 + l];
                        y[IDX5(n, co, d, h, ww, Cout, D2, H2, W2)] = (float)acc;
                    }
}
void orc_convT2_bwd(const float *x, const float *w, const float *gy, float *gx, float *gw, float *gb,
                    int N, int Cin, int Cout, int D, int H, int W) {
    int D2 = 2 * D, H2 = 2 * H, W2 = 2 * W;
    size_t nx = (size_t)N * Cin * D * H * W, nw = (size_t)Cin * Cout * 8;
    double *dgx = (double *)calloc(nx, sizeof(double)), *dgw = (double *)calloc(nw, sizeof(double));
    for (int co = 0; co < Cout; co++) {
        double sb = 0.0;
        for (int n = 0; n < N; n++)
            for (int d = 0; d < D2; d++)
                for (int h = 0; h < H2; h++)
                    for (int ww = 0; ww < W2; ww++) {
                        double g = gy[IDX5(n, co, d, h, ww, Cout, D2, H2, W2)];
                        sb += g;
                        int i = d & 1, j = h & 1, l = ww & 1;
                        for (int ci = 0; ci < Cin; ci++) {
                            size_t xi = IDX5(n, ci, d / 2, h / 2, ww / 2, Cin, D, H, W);
                            size_t wi = ((((size_t)ci * Cout + co) * 2 + i) * 2 + j) * 2 + l;
                            dgx[xi] += g * (double)w[wi];
                            dgw[wi] += g * (double)x[xi];
                        }
                    }
        if (gb) gb[co] = (float)sb;
    }
    if (gx) for (size_t i = 0; i < nx; i++) gx[i] = (float)dgx[i];
    if (gw) for (size_t i = 0; i < nw; i++) gw[i] = (float)dgw[i];
    free(dgx);
    free(dgw);
}

/* Segmentation loss family, utils/metrics.py:14-40 (combined_loss), :137-156 (tversky_loss),
 * :158-167 (combined_ce_tversky_loss), :169-190 (distillation_loss); train_unet.py:186-198 ('dice').
 *   loss = w_ce * CE_mean + w_reg * mean_{c=1..C-1} region_c + w_kd * T^2 * mean_{n,c,v} KL
 *   region kind 1 (dice):    1 - (2 I_c + eps) / (P_c + T_c + eps)
 *   region kind 2 (tversky): 1 - (I_c + eps) / (I_c + a*FP_c + b*FN_c + eps), FP = sum p(1-t), FN = sum (1-p)t
 * grad may be NULL; teacher may be NULL when w_kd == 0.  sums_out (optional) = [CE_sum, I_c, P_c, T_c ...]. */
void orc_seg_loss(const float *logits, const int64_t *labels, const float *teacher,
                  int N, int C, int64_t V, double w_ce, int region_kind, double w_reg,
                  double alpha, double beta, double eps, double w_kd, double temp,
                  double *loss_out, float *grad) {
    double M = (double)N * (double)V;
    double *I = (double *)calloc(C, sizeof(double)), *P = (double *)calloc(C, sizeof(double));
    double *Tn = (double *)calloc(C, sizeof(double)), *FPs = (double *)calloc(C, sizeof(double));
    double *FNs = (double *)calloc(C, sizeof(double));
    double *p = (double *)malloc(sizeof(double) * C), *ps = (double *)malloc(sizeof(double) * C);
    double *pt = (double *)malloc(sizeof(double) * C);
    double ce = 0.0, kl = 0.0;
    for (int n = 0; n < N; n++)
        for (int64_t v = 0; v < V; v++) {
            int64_t t = labels[(size_t)n * V + v];
            double mx = -INFINITY, se = 0.0;
            for (int c = 0; c < C; c++) { double z = logits[((size_t)n * C + c) * V + v]; if (z > mx) mx = z; }
            for (int c = 0; c < C; c++) { p[c] = exp(logits[((size_t)n * C + c) * V + v] - mx); se += p[c]; }
            for (int c = 0; c < C; c++) p[c] /= se;
            ce += -(logits[((size_t)n * C + t) * V + v] - mx - log(se));
            for (int c = 0; c < C; c++) {
                double tc = (t == c) ? 1.0 : 0.0;
                I[c] += p[c] * tc; P[c] += p[c]; Tn[c] += tc;
                FPs[c] += p[c] * (1.0 - tc); FNs[c] += (1.0 - p[c]) * tc;
            }
            if (w_kd != 0.0) {
                double ms = -INFINITY, mt = -INFINITY, ss = 0.0, st = 0.0;
                for (int c = 0; c < C; c++) {
                    double a = logits[((size_t)n * C + c) * V + v] / temp, b = teacher[((size_t)n * C + c) * V + v] / temp;
                    if (a > ms) ms = a;
                    if (b > mt) mt = b;
                }
                for (int c = 0; c < C; c++) {
                    ss += exp(logits[((size_t)n * C + c) * V + v] / temp - ms);
                    st += exp(teacher[((size_t)n * C + c) * V + v] / temp - mt);
                }
                for (int c = 0; c < C; c++) {
                    double ls = logits[((size_t)n * C + c) * V + v] / temp - ms - log(ss);
                    double lt = teacher[((size_t)n * C + c) * V + v] / temp - mt - log(st);
                    double q = exp(lt);
                    if (q > 0.0) kl += q * (lt - ls);   /* F.kl_div(log_q_student, p_teacher): xlogy semantics */
                }
            }
        }
    double reg = 0.0;
    double *A = (double *)calloc(C, sizeof(double)), *B = (double *)calloc(C, sizeof(double));
    /* dRegion/dp_c at a voxel = A_c * [t=c] + B_c */
    for (int c = 1; c < C; c++) {
        if (region_kind == 1) {
            double U = P[c] + Tn[c];
            reg += 1.0 - (2.0 * I[c] + eps) / (U + eps);
            A[c] = -2.0 / (U + eps);
            B[c] = (2.0 * I[c] + eps) / ((U + eps) * (U + eps));
        } else if (region_kind == 2) {
            double num = I[c] + eps, den = I[c] + alpha * FPs[c] + beta * FNs[c] + eps;
            reg += 1.0 - num / den;
            /* d num/dp = t ; d den/dp = t + alpha(1-t) - beta t */
            A[c] = -(1.0 / den) + num / (den * den) * (1.0 - alpha - beta);
            B[c] = num / (den * den) * alpha;
        }
    }
    if (C > 1) reg /= (double)(C - 1);
    double loss = w_ce * ce / M + w_reg * reg + w_kd * temp * temp * kl / (M * C);
    if (loss_out) *loss_out = loss;
    if (grad) {
        for (int n = 0; n < N; n++)
            for (int64_t v = 0; v < V; v++) {
                int64_t t = labels[(size_t)n * V + v];
                double mx = -INFINITY, se = 0.0;
                for (int c = 0; c < C; c++) { double z = logits[((size_t)n * C + c) * V + v]; if (z > mx) mx = z; }
                for (int c = 0; c < C; c++) { p[c] = exp(logits[((size_t)n * C + c) * V + v] - mx); se += p[c]; }
                double dot = 0.0;
                for (int c = 0; c < C; c++) {
                    p[c] /= se;
                    double g = (c >= 1 && C > 1) ? w_reg / (double)(C - 1) * (A[c] * ((t == c) ? 1.0 : 0.0) + B[c]) : 0.0;
                    pt[c] = g;
                    dot += g * p[c];
                }
                if (w_kd != 0.0) {
                    double ms = -INFINITY, mt = -INFINITY, ss = 0.0, st = 0.0;
                    for (int c = 0; c < C; c++) {
                        double a = logits[((size_t)n * C + c) * V + v] / temp, b = teacher[((size_t)n * C + c) * V + v] / temp;
                        if (a > ms) ms = a;
                        if (b > mt) mt = b;
                    }
                    for (int c = 0; c < C; c++) {
                        ps[c] = exp(logits[((size_t)n * C + c) * V + v] / temp - ms); ss += ps[c];
                    }
                    for (int c = 0; c < C; c++) ps[c] /= ss;
                    for (int c = 0; c < C; c++) st += exp(teacher[((size_t)n * C + c) * V + v] / temp - mt);
                    for (int c = 0; c < C; c++) {
                        double q = exp(teacher[((size_t)n * C + c) * V + v] / temp - mt) / st;
                        ps[c] = w_kd * temp * (ps[c] - q) / (M * C);
                    }
                }
                for (int c = 0; c < C; c++) {
                    double g = w_ce * (p[c] - ((t == c) ? 1.0 : 0.0)) / M + p[c] * (pt[c] - dot);
                    if (w_kd != 0.0) g += ps[c];
                    grad[((size_t)n * C + c) * V + v] = (float)g;
                }
            }
    }
    free(I); free(P); free(Tn); free(FPs); free(FNs); free(p); free(ps); free(pt); free(A); free(B);
}

/* calculate_iou / calculate_dice / calculate_accuracy: utils/metrics.py:65-129.
 * Q1: the class loop runs range(1, D) AFTER argmax (D = first spatial dim), and a class is
 * scored only if present in the target.  argmax ties resolve to the lowest index (torch.argmax).
 * out = {iou, dice, acc}; counts (optional, 3*C+1 int64) = n_inter[c], n_pred[c], n_tgt[c], n_correct. */
void orc_seg_metrics(const float *logits, const int64_t *labels, int N, int C, int D, int64_t V,
                     double *out, int64_t *counts) {
    int64_t *ni = (int64_t *)calloc(C, sizeof(int64_t)), *np_ = (int64_t *)calloc(C, sizeof(int64_t));
    int64_t *nt = (int64_t *)calloc(C, sizeof(int64_t)), correct = 0;
    for (int n = 0; n < N; n++)
        for (int64_t v = 0; v < V; v++) {
            int best = 0;
            float bv = logits[((size_t)n * C) * V + v];
            for (int c = 1; c < C; c++) {
                float z = logits[((size_t)n * C + c) * V + v];
                if (z > bv) { bv = z; best = c; }
            }
            int64_t t = labels[(size_t)n * V + v];
            np_[best]++;
            if (t >= 0 && t < C) nt[t]++;
            if (best == t) { ni[best]++; correct++; }
        }
    double iou = 0.0, dice = 0.0;
    int valid = 0;
    for (int c = 1; c < D && c < C; c++) {
        if (nt[c] > 0) {
            /* reference computes these ratios in float32 tensors */
            float inter = (float)ni[c];
            float uni = (float)(np_[c] + nt[c]) - inter;
            iou += (double)((inter + 1e-5f) / (uni + 1e-5f));
            dice += (double)((2.f * inter + 1e-5f) / ((float)(np_[c] + nt[c]) + 1e-5f));
            valid++;
        }
    }
    int dv = valid > 1 ? valid : 1;
    out[0] = iou / dv;
    out[1] = dice / dv;
    out[2] = (double)correct / ((double)N * (double)V);
    if (counts) {
        for (int c = 0; c < C; c++) { counts[c] = ni[c]; counts[C + c] = np_[c]; counts[2 * C + c] = nt[c]; }
        counts[3 * C] = correct;
    }
    free(ni); free(np_); free(nt);
}

/* torch.mean(bottleneck, dim=[2,3,4]): models/unet_dann.py:79 */
void orc_gap_fwd(const float *x, float *y, int N, int C, int64_t V) {
    for (size_t nc = 0; nc < (size_t)N * C; nc++) {
        double s = 0.0;
        for (int64_t v = 0; v < V; v++) s += x[nc * V + v];
        y[nc] = (float)(s / (double)V);
    }
}

/* nn.Linear: train_dann.py:38-46.  y = x W^T + b, W (out,in). */
void orc_linear_fwd(const float *x, const float *w, const float *b, float *y, int M, int K, int Nout) {
    for (int m = 0; m < M; m++)
        for (int o = 0; o < Nout; o++) {
            double a = b ? b[o] : 0.0;
            for (int k = 0; k < K; k++) a += (double)x[(size_t)m * K + k] * (double)w[(size_t)o * K + k];
            y[(size_t)m * Nout + o] = (float)a;
        }
}
void orc_linear_bwd(const float *x, const float *w, const float *gy, float *gx, float *gw, float *gb,
                    int M, int K, int Nout) {
    for (int m = 0; m < M; m++)
        for (int k = 0; k < K; k++) {
            double a = 0.0;
            for (int o = 0; o < Nout; o++) a += (double)gy[(size_t)m * Nout + o] * (double)w[(size_t)o * K + k];
            gx[(size_t)m * K + k] = (float)a;
        }
    for (int o = 0; o < Nout; o++) {
        double sb = 0.0;
        for (int m = 0; m < M; m++) sb += gy[(size_t)m * Nout + o];
        if (gb) gb[o] = (float)sb;
        for (int k = 0; k < K; k++) {
            double a = 0.0;
            for (int m = 0; m < M; m++) a += (double)gy[(size_t)m * Nout + o] * (double)x[(size_t)m * K + k];
            gw[(size_t)o * K + k] = (float)a;
        }
    }
}

/* torch.optim.AdamW single-tensor step (train_unet.py:378): decoupled weight decay, bias correction. */
void orc_adamw_step(float *p, const float *g, float *m, float *v, int64_t n, double lr, double b1,
                    double b2, double eps, double wd, int64_t step) {
    double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    for (int64_t i = 0; i < n; i++) {
        double pi = p[i] * (1.0 - lr * wd);
        double mi = b1 * m[i] + (1.0 - b1) * g[i];
        double vi = b2 * v[i] + (1.0 - b2) * (double)g[i] * g[i];
        m[i] = (float)mi;
        v[i] = (float)vi;
        p[i] = (float)(pi - lr / bc1 * mi / (sqrt(vi) / sqrt(bc2) + eps));
    }
}
