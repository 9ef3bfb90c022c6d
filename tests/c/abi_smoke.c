/* Compiled as C99 by tests/test_abi.py: the public header must be plain C, and a C caller must be able to load a model,
 * fail cleanly without a GPU (or run one frame with one), and read the error string.  Mirrors what a maintainer of the
 * reference would write against include/percepnet_hip.h (level 2). */
#include "percepnet_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s model.pnw\n", argv[0]); return 2; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror("model"); return 2; }
  pn_model *m = pn_model_from_file(f);
  fclose(f);
  if (!m) { fprintf(stderr, "model: %s\n", pn_last_error()); return 3; }
  printf("version %s, %d kernel families, frame %d samples, %d bands, %d features\n", pn_version(), pn_kernel_count(),
         PN_FRAME_SIZE, PN_NB_BANDS, PN_NB_FEATURES);
  pn_ctx *cx = pn_ctx_create(m, 0, 3, PN_NN_MFMA, NULL);
  if (!cx) {
    printf("no context: %s\n", pn_last_error());        /* expected on a machine without a HIP device */
    pn_model_free(m);
    return 0;
  }
  {
    static int16_t in[3 * PN_FRAME_SIZE], out[3 * PN_FRAME_SIZE];
    static float gr[3 * 68];
    int i, rc;
    for (i = 0; i < 3 * PN_FRAME_SIZE; i++) in[i] = (int16_t)((i * 37) % 2001 - 1000);
    rc = pn_process_host_i16(cx, in, out, gr);
    printf("process rc=%d streams=%d frames=%lld bytes=%zu g0=%f\n", rc, pn_ctx_n_streams(cx), (long long)pn_ctx_frames_done(cx),
           pn_ctx_device_bytes(cx), gr[0]);
    pn_ctx_destroy(cx);
    pn_model_free(m);
    return rc ? 4 : 0;
  }
}
