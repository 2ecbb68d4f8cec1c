/* A C99 caller of include/ptrace.h -- the boundary is a C ABI, not a Python one.
 *   probe sizes            -> the struct sizes and the version the library reports (no GPU needed)
 *   probe render W H       -> uploads a two-shape scene (a sphere in front of a checkered plane), renders it with the
 *                             FlatRenderer parameters through pt_render and prints every pixel as hex doubles
 * Built and run by tests/test_c_abi.py (gcc -std=c99 -Wall -Werror -pedantic). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ptrace.h"

static int fail(const char *what, int rc) {
  char msg[512];
  pt_last_error(msg, sizeof msg);
  fprintf(stderr, "%s failed (%d): %s\n", what, rc, msg);
  return 1;
}

int main(int argc, char **argv) {
  if (argc >= 2 && strcmp(argv[1], "sizes") == 0) {
    pt_params p;
    memset(&p, 0, sizeof p);
    p.width = 7;
    p.height = 21;
    p.row_block = 8;
    p.n_ranks = 2;
    p.rank = 1;
    p.out_format = PT_OUT_F32;
    printf("version %d\n", pt_version());
    printf("sizeof pt_scene_desc %zu pt_camera %zu pt_params %zu pt_stats %zu\n", sizeof(pt_scene_desc), sizeof(pt_camera),
           sizeof(pt_params), sizeof(pt_stats));
    printf("rows_for_rank %d output_bytes %zu sparse_fixed %lld\n", pt_rows_for_rank(&p), pt_output_bytes(&p),
           pt_image_sparse_fixed_bytes(1000, PT_OUT_F32));
    return 0;
  }
  if (argc >= 4 && strcmp(argv[1], "render") == 0) {
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    /* two shapes, structure-of-arrays: element (r, c) of shape i at a[(r * 4 + c) * n + i] */
    enum { N = 2 };
    int32_t kind[N] = {PT_SHAPE_SPHERE, PT_SHAPE_PLANE};
    double invm[12 * N], m[12 * N];
    memset(invm, 0, sizeof invm);
    memset(m, 0, sizeof m);
    /* sphere: translation(2, 0.25, 0.5) * scaling(0.5); plane: identity */
    const double s = 0.5, t[3] = {2.0, 0.25, 0.5};
    for (int r = 0; r < 3; ++r) {
      m[(r * 4 + r) * N + 0] = s;
      m[(r * 4 + 3) * N + 0] = t[r];
      invm[(r * 4 + r) * N + 0] = 1.0 / s;
      invm[(r * 4 + 3) * N + 0] = -t[r] / s;
      m[(r * 4 + r) * N + 1] = 1.0;
      invm[(r * 4 + r) * N + 1] = 1.0;
    }
    int32_t brdf_kind[N] = {PT_BRDF_DIFFUSE, PT_BRDF_DIFFUSE}, pig_kind[N] = {PT_PIGMENT_UNIFORM, PT_PIGMENT_CHECKERED};
    int32_t emi_kind[N] = {PT_PIGMENT_UNIFORM, PT_PIGMENT_UNIFORM}, tex[N] = {-1, -1};
    double brdf_param[N] = {0.0, 0.0}, steps[N] = {1.0, 4.0}, esteps[N] = {1.0, 1.0};
    /* colours are channel-major: channel k of shape i at c[k * n + i] */
    double pig_c1[3 * N] = {0.9, 0.5, 0.3, 0.1, 0.2, 0.1}, pig_c2[3 * N] = {0.0, 0.2, 0.0, 0.0, 0.0, 0.5};
    double emi_c1[3 * N] = {0.125, 0.0, 0.0, 0.0, 0.0, 0.0}, emi_c2[3 * N] = {0, 0, 0, 0, 0, 0};
    pt_scene_desc d;
    memset(&d, 0, sizeof d);
    d.n_shapes = N;
    d.kind = kind;
    d.invm = invm;
    d.m = m;
    d.brdf_kind = brdf_kind;
    d.brdf_param = brdf_param;
    d.pig_kind = pig_kind;
    d.pig_c1 = pig_c1;
    d.pig_c2 = pig_c2;
    d.pig_steps = steps;
    d.pig_tex = tex;
    d.emi_kind = emi_kind;
    d.emi_c1 = emi_c1;
    d.emi_c2 = emi_c2;
    d.emi_steps = esteps;
    d.emi_tex = tex;
    pt_camera cam;
    memset(&cam, 0, sizeof cam);
    cam.kind = PT_CAMERA_PERSPECTIVE;
    cam.m[0] = cam.m[5] = cam.m[10] = 1.0;
    cam.m[3] = -1.0;
    cam.m[11] = 1.0; /* translation(-1, 0, 1) */
    cam.screen_distance = 1.0;
    cam.aspect_ratio = (double)W / (double)H;
    pt_params p;
    memset(&p, 0, sizeof p);
    p.width = W;
    p.height = H;
    p.renderer = PT_RENDERER_FLAT;
    p.background[2] = 0.25;
    p.num_of_rays = 1;
    p.pcg_mode = PT_PCG_PIXEL;
    p.row_block = 8;
    p.n_ranks = 1;
    p.out_format = PT_OUT_F64;
    pt_scene *scene = NULL;
    int rc = pt_scene_upload(&d, 0, &scene);
    if (rc) return fail("pt_scene_upload", rc);
    const size_t bytes = pt_output_bytes(&p);
    double *out = (double *)malloc(bytes);
    rc = pt_render(scene, &cam, &p, out, bytes);
    if (rc) return fail("pt_render", rc);
    pt_stats st;
    rc = pt_get_stats(scene, &st);
    if (rc) return fail("pt_get_stats", rc);
    printf("rays %llu pixels %llu kernel %d\n", (unsigned long long)st.n_rays, (unsigned long long)st.n_pixels, st.kernel);
    for (size_t i = 0; i < (size_t)W * H * 3; ++i) printf("%a\n", out[i]);
    /* the same frame left in HBM: buffer and stream from the C-ABI itself (1.5), no GPU framework on the caller's side */
    void *dev = NULL, *stream = NULL;
    double *back = (double *)malloc(bytes);
    rc = pt_device_alloc(0, bytes, &dev);
    if (rc) return fail("pt_device_alloc", rc);
    rc = pt_stream_create(0, &stream);
    if (rc) return fail("pt_stream_create", rc);
    rc = pt_render_device(scene, &cam, &p, dev, bytes, stream);
    if (rc) return fail("pt_render_device", rc);
    rc = pt_stream_sync(0, stream);
    if (rc) return fail("pt_stream_sync", rc);
    rc = pt_device_download(0, back, dev, bytes, stream);
    if (rc) return fail("pt_device_download", rc);
    printf("device frame identical %d\n", memcmp(out, back, bytes) == 0);
    rc = pt_stream_destroy(0, stream);
    if (rc) return fail("pt_stream_destroy", rc);
    rc = pt_device_free(0, dev);
    if (rc) return fail("pt_device_free", rc);
    printf("free(NULL) %d %d alloc(0) %d\n", pt_device_free(0, NULL), pt_stream_destroy(0, NULL), pt_device_alloc(0, 0, &dev) == PT_OK && dev == NULL);
    printf("bad device -> %d\n", pt_device_alloc(4096, 16, &dev));
    free(back);
    rc = pt_render(scene, &cam, &p, out, bytes - 1); /* a buffer one byte short must be refused, not overrun */
    printf("short buffer -> %d\n", rc);
    free(out);
    pt_scene_free(scene);
    return 0;
  }
  fprintf(stderr, "usage: probe sizes | probe render W H\n");
  return 2;
}
