/* Plain-C client of include/rt_tile.h: what a cgo / Rust FFI user does, without Python.
 * Renders the c1-style scene (one sphere) as `divisions` strips over all GPUs with rt_render_frame,
 * cross-checks strip 3 against rt_render_tile and two frames of a persistent rt_frame_ctx against the first, writes a
 * binary PPM.
 *   gcc -std=c99 -O2 -Iinclude examples/render_frame.c -Lray_tracer_s8_amd/lib -lrt_s8 -Wl,-rpath,... -o render_frame
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_tile.h"

int main(int argc, char** argv) {
    const char* out_path = argc > 1 ? argv[1] : "frame.ppm";
    int n_dev = 0;
    int rc = rt_init(&n_dev);
    if (rc != RT_OK) {
        fprintf(stderr, "rt_init: %s (%s)\n", rt_strerror(rc), rt_last_error());
        return 2;                       /* no GPU: fail loudly, there is no CPU fallback */
    }
    rt_sphere world[2] = {
        {0.0f, 0.0f, -3.0f, 1.0f, 0.8f, 0.3f, 0.3f, 0.0f, 0.0f},
        {0.0f, -101.0f, -3.0f, 100.0f, 0.5f, 0.5f, 0.5f, 0.0f, 0.0f},
    };
    rt_tile_request rq;
    rt_tile_request_defaults(&rq);      /* reference literals: 100 spp, 10 bounces, camera */
    rq.width = 256;
    rq.height = 160;
    rq.divisions = 8;
    rq.spp = 16;
    rq.seed = 42;
    const size_t frame_bytes = (size_t)rq.width * rq.height * 3;
    unsigned char* frame = (unsigned char*)malloc(frame_bytes);
    rt_tile_stats st;
    rc = rt_render_frame(NULL, 0, &rq, world, 2, NULL, 0, NULL, frame, frame_bytes, &st);
    if (rc != RT_OK) {
        fprintf(stderr, "rt_render_frame: %s (%s)\n", rt_strerror(rc), rt_last_error());
        return 1;
    }
    /* one strip through the slave-style entry point must equal the same rows of the frame */
    const size_t strip_bytes = rt_tile_bytes(&rq);
    unsigned char* strip = (unsigned char*)malloc(strip_bytes);
    rq.division_no = 3;
    rc = rt_render_tile(0, &rq, world, 2, NULL, 0, NULL, strip, strip_bytes, NULL, NULL);
    if (rc != RT_OK || memcmp(strip, frame + 3 * strip_bytes, strip_bytes) != 0) {
        fprintf(stderr, "strip 3 differs from the frame (rc=%d)\n", rc);
        return 1;
    }
    /* the same job through a persistent frame context: two frames, the second pays no pinning and no world upload */
    {
        rt_frame_ctx* job = NULL;
        rt_frame_stats fs;
        unsigned char* frame2 = (unsigned char*)malloc(frame_bytes);
        rc = rt_frame_ctx_create(NULL, 0, &job);
        if (rc == RT_OK) rc = rt_frame_ctx_set_world(job, world, 2, NULL, 0, NULL);
        for (int i = 0; i < 2 && rc == RT_OK; i++) {
            rc = rt_frame_ctx_render(job, &rq, frame2, frame_bytes, &fs);
            if (rc == RT_OK && memcmp(frame2, frame, frame_bytes) != 0) rc = -100;
        }
        if (rc == RT_OK && (fs.pin_ms != 0.0f || fs.scene_ms != 0.0f)) rc = -101;
        rt_frame_ctx_destroy(job);          /* before the buffer it has page-locked is freed */
        free(frame2);
        if (rc != RT_OK) {
            fprintf(stderr, "frame context: rc=%d (%s)\n", rc, rt_last_error());
            return 1;
        }
    }
    FILE* f = fopen(out_path, "wb");
    if (!f) return 1;
    fprintf(f, "P6\n%u %u\n255\n", rq.width, rq.height);
    fwrite(frame, 1, frame_bytes, f);
    fclose(f);
    printf("C_CLIENT_OK devices=%d segments=%llu kernel_ms=%.3f launches=%u -> %s\n", n_dev,
           (unsigned long long)st.ray_segments, st.kernel_ms, st.n_launches, out_path);
    free(strip);
    free(frame);
    rt_shutdown();
    return 0;
}
