#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "gsplat_image.h"
int main(int argc, char **argv) {
    long total = 0, okc = 0;
    for (int a = 1; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb"); if (!f) continue;
        uint8_t *blob = malloc(1 << 20); size_t n = fread(blob, 1, 1 << 20, f); fclose(f);
        srand(a);
        for (int it = 0; it < 4000; it++) {
            size_t m = n;
            uint8_t *b = malloc(n); memcpy(b, blob, n);
            int k = 1 + rand() % 6;
            for (int i = 0; i < k; i++) b[2 + rand() % (n - 2)] = (uint8_t)rand();
            if (rand() % 5 == 0) m = 10 + rand() % (n - 10);
            int w = 0, h = 0, c = 0;
            int rc = gs_jpeg_info(b, m, &w, &h, &c);
            if (rc == 0 && w > 0 && h > 0 && (long)w * h <= 4096L * 4096L) {
                uint8_t *out = malloc((size_t)w * h * 3);
                rc = gs_jpeg_decode_rgb(b, m, out, (size_t)w * h * 3);
                if (rc == 0) okc++;
                free(out);
            }
            free(b); total++;
        }
        free(blob);
    }
    printf("%ld mutated files, %ld decoded\n", total, okc);
    return 0;
}
