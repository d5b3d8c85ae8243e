/* image_host.c -- a host WITHOUT Python (and without any HIP binding of its own) running a compiled U-Net forward through
 * the C ABI of libaed.so (include/aed.h, "tape images").  What the Python side did ahead of time:
 *     audioeditingcode_amd.image.export_image("unet.aedimg", {"context": eng.ctx_tape, "forward": eng.tape}, {...names...})
 * What this program does: load the image, optionally overwrite named inputs from raw little-endian files, run the
 * per-prompt "context" program once and the "forward" program n times, write the named output to a file.
 *
 *   gcc -O2 -Iinclude examples/image_host.c -Laudioeditingcode_amd -laed -Wl,-rpath,'$ORIGIN' -o audioeditingcode_amd/aed_image_host
 *   aed_image_host unet.aedimg eps.bin [n_forwards] [name=file.bin ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "aed.h"

#define CHECK(call)                                                            \
    do {                                                                       \
        int rc_ = (call);                                                      \
        if (rc_) {                                                             \
            fprintf(stderr, "%s failed (rc=%d): %s\n", #call, rc_, aed_last_error()); \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s image.aedimg out.bin [n_forwards] [name=file.bin ...]\n", argv[0]);
        return 2;
    }
    int n_forwards = argc > 3 ? atoi(argv[3]) : 1;
    void *image = NULL, *stream = NULL, *ev0 = NULL, *ev1 = NULL;
    CHECK(aed_image_load(argv[1], 0, &image));
    CHECK(aed_stream_create_cu_mask(&stream, NULL, 0, 0));          /* a plain (unmasked) stream owned by the library */
    for (int a = 4; a < argc; ++a) {                                 /* named inputs from files */
        char* eq = strchr(argv[a], '=');
        if (!eq) { fprintf(stderr, "bad argument %s (want name=file)\n", argv[a]); return 2; }
        *eq = 0;
        FILE* f = fopen(eq + 1, "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", eq + 1); return 1; }
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        void* buf = malloc((size_t)n);
        if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read on %s\n", eq + 1); return 1; }
        fclose(f);
        CHECK(aed_image_copy_in(image, argv[a], buf, (uint64_t)n, stream));
        free(buf);
    }
    CHECK(aed_image_run(image, "context", stream));                  /* cross-attention K/V of the prompt: once */
    CHECK(aed_event_create(&ev0));
    CHECK(aed_event_create(&ev1));
    CHECK(aed_event_record(ev0, stream));
    for (int k = 0; k < n_forwards; ++k) CHECK(aed_image_run(image, "forward", stream));
    CHECK(aed_event_record(ev1, stream));
    float ms = 0.f;
    CHECK(aed_event_elapsed_ms(ev0, ev1, &ms));
    void* dev = NULL;
    uint64_t nbytes = 0;
    CHECK(aed_image_buffer(image, "eps", &dev, &nbytes));
    void* out = malloc((size_t)nbytes);
    CHECK(aed_image_copy_out(image, "eps", out, nbytes, stream));
    FILE* f = fopen(argv[2], "wb");
    if (!f || fwrite(out, 1, (size_t)nbytes, f) != (size_t)nbytes) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
    fclose(f);
    free(out);
    printf("%d forward(s) in %.3f ms (%.3f ms each); %llu bytes of eps -> %s\n", n_forwards, ms, ms / n_forwards,
           (unsigned long long)nbytes, argv[2]);
    CHECK(aed_event_destroy(ev0));
    CHECK(aed_event_destroy(ev1));
    CHECK(aed_stream_destroy(stream));
    CHECK(aed_image_free(image));
    return 0;
}
