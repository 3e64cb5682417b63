/*
 * o_formats.c -- oracle: the consumer right after fft.receive: one waterfall pixel row per PSD frame.
 * TEST INFRASTRUCTURE (see jsdr_oracle.h).  Build with -ffp-contract=off.
 */
#include "jsdr_oracle.h"

/* Java (int) of a float: truncate toward zero, saturate, NaN -> 0 */
static int java_f2i(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

/* waterfall.java:102-109 getMax: largest value from offset o, length l (l <= 0 gives a[o]); '>' never
 * replaces with or by a NaN */
static float wf_getmax(const float *a, int o, int l)
{
    float r = a[o];
    for (int i = o + 1; i < o + l; i++)
        if (a[i] > r) r = a[i];
    return r;
}

/* waterfall.java:60-61,87-100 paintLine: psd has n+2 entries (n bins, then Hz and the maximum, fft.java:226-227);
 * step = (float)n/(float)width, pixel p shows the maximum of bins [(int)(p*step), +(int)step), mapped
 * -100 dBFS -> 255 ... 0 dBFS -> 0 by 255-(int)(m*-2.55f), clamped, scaled into the peak colour with integer
 * division by 256, and stored at column (p + width/2) % width as ARGB with alpha 0xff (Color.getRGB()).  */
void jo_waterfall_line(const float *psd, int n, int width, unsigned peak_rgb, unsigned *pix)
{
    const float step = (float)n / (float)width;
    const int off = width / 2;
    const float h = -2.55f;
    const int pr = (peak_rgb >> 16) & 0xff, pg = (peak_rgb >> 8) & 0xff, pb = peak_rgb & 0xff;
    for (int p = 0; p < width; p++) {
        int f = 255 - java_f2i(wf_getmax(psd, java_f2i((float)p * step), java_f2i(step)) * h);
        f = f < 0 ? 0 : f;
        f = f > 255 ? 255 : f;
        const unsigned r = (unsigned)(pr * f / 256), g = (unsigned)(pg * f / 256), b = (unsigned)(pb * f / 256);
        pix[(p + off) % width] = 0xff000000u | (r << 16) | (g << 8) | b;
    }
}
