/*
 * A plain C program on the drop-in boundary (include/libvmaf.h), the way a C host such as FFmpeg's libvmaf filter sits on
 * the reference's wrapper/src/libvmaf.h:26: it owns the frames, the library pulls them through read_frame.
 * Built with gcc (no C++, no HIP headers) by tests/test_gpu_cli.py and linked against vmaf_amd/libvmaf_amd.so.
 *
 *   c_caller ref.yuv dis.yuv width height model.pkl [fail_at_frame]
 *
 * 8-bit yuv420p input.  Prints "score <pooled VMAF>" or "error <message>" and exits 0 / 3.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "libvmaf.h"

struct clip { FILE *ref, *dis; int w, h, frame, fail_at; unsigned char *row; };

static int read_plane(FILE *f, struct clip *c, float *dst, int stride_byte)
{
    int i, j;
    for (i = 0; i < c->h; ++i) {
        float *out = (float *)((char *)dst + (size_t)i * stride_byte);
        if (fread(c->row, 1, (size_t)c->w, f) != (size_t)c->w) return 1;
        for (j = 0; j < c->w; ++j) out[j] = (float)c->row[j];
    }
    return fseek(f, (long)c->w * c->h / 2, SEEK_CUR) != 0;          /* skip U and V */
}

/* the reference's read_frame contract: 0 = frame delivered, 2 = end of stream, 1 = failure (combo.c:149-157) */
static int read_frame(float *ref_data, float *main_data, float *temp_data, int stride_byte, void *user_data)
{
    struct clip *c = (struct clip *)user_data;
    (void)temp_data;
    if (c->frame == c->fail_at) return 1;
    if (read_plane(c->ref, c, ref_data, stride_byte)) return 2;
    if (read_plane(c->dis, c, main_data, stride_byte)) return 2;
    ++c->frame;
    return 0;
}

int main(int argc, char **argv)
{
    struct clip c;
    double score = 0.0;
    char err[512] = "";
    char fmt[] = "yuv420p";
    int rc;
    if (argc < 6) { fprintf(stderr, "usage: c_caller ref dis w h model [fail_at_frame]\n"); return 2; }
    memset(&c, 0, sizeof c);
    c.w = atoi(argv[3]);
    c.h = atoi(argv[4]);
    c.fail_at = argc > 6 ? atoi(argv[6]) : -1;
    c.ref = fopen(argv[1], "rb");
    c.dis = fopen(argv[2], "rb");
    c.row = (unsigned char *)malloc((size_t)c.w);
    if (!c.ref || !c.dis || !c.row) { fprintf(stderr, "cannot open inputs\n"); return 2; }
    rc = compute_vmaf_noexcept(&score, fmt, c.w, c.h, read_frame, &c, argv[5], NULL, NULL, 0, 0, 0, 0, 0, 0, 0, NULL, err, (int)sizeof err);
    if (rc) printf("error %s\n", err);
    else printf("score %.6f\n", score);
    fclose(c.ref);
    fclose(c.dis);
    free(c.row);
    return rc ? 3 : 0;
}
