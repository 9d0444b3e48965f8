// Image file decoding for glTF textures.  The reference decodes with stb_image forced to 4 channels
// (src/gltf.cpp:103-112: stbi_load(path, &w, &h, &channels, 4)); this is an independent decoder with the same
// contract: 8-bit RGBA, row-major from the top-left, no vertical flip.
//   PNG  (all colour types, bit depths 1-16, tRNS, Adam7): lossless, so the texels are defined by the file.
//   JPEG (baseline, extended sequential and progressive Huffman, 8-bit, 1 or 3 components, any sampling factors,
//        interleaved and per-component scans, restart intervals): the standard leaves IDCT rounding and chroma upsampling to the decoder, texels can differ
//        by a few code values from another decoder's.
#ifndef PTR_IMAGE_HPP_
#define PTR_IMAGE_HPP_

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

struct image_error : std::runtime_error { using std::runtime_error::runtime_error; };

struct raw_image {
    int32_t width = 0;
    int32_t height = 0;
    int32_t channels = 0;       // channels the FILE stores (1..4); `data` is always 4 per texel
    std::vector<uint8_t> data;  // width * height * 4, RGBA8
};

// RFC 1950 / 1951: zlib stream -> bytes.  `expected_size` (0 = unknown) bounds the output.
std::vector<uint8_t> zlib_inflate(const uint8_t* src, size_t size, size_t expected_size = 0);

// RGB8 PNG file image (colour type 2, stored deflate blocks, filter 0): lossless and readable everywhere; nothing is compressed.
std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb, uint32_t width, uint32_t height);

raw_image decode_png(const uint8_t* bytes, size_t size);
raw_image decode_jpeg(const uint8_t* bytes, size_t size);
raw_image decode_image(const uint8_t* bytes, size_t size);  // sniffs the signature
raw_image load_image(const std::string& path);

#endif  // PTR_IMAGE_HPP_
