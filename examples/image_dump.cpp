// rt_image — loads an image the way an image_texture would (rtw_image: JPEG, PPM or a raw
// texel dump) and writes the linear 8-bit texels as "RTIMG1\nW H\n" + W*H*3 bytes.
//   usage: rt_image <in.jpg|.ppm|.rtimg> <out.rtimg>
#include "rtw_stb_image.h"

#include <cstdio>

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: rt_image <in> <out.rtimg>\n");
        return 2;
    }
    rtw_image im;
    if (!im.load(argv[1])) {
        fprintf(stderr, "rt_image: cannot load %s\n", argv[1]);
        return 1;
    }
    FILE *f = fopen(argv[2], "wb");
    if (!f) return 1;
    fprintf(f, "RTIMG1\n%d %d\n", im.width(), im.height());
    fwrite(im.bytes().data(), 1, im.bytes().size(), f);
    fclose(f);
    printf("%d x %d\n", im.width(), im.height());
    return 0;
}
