#include "HdrImage.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace MRendererHip {

namespace {

// next text line of the header (without the terminating '\n'); false at end of buffer
bool NextLine(const uint8_t* file, size_t bytes, size_t& pos, std::string& line) {
    if (pos >= bytes) return false;
    size_t end = pos;
    while (end < bytes && file[end] != '\n') end++;
    if (end == bytes) return false;   // header lines are newline-terminated
    line.assign(reinterpret_cast<const char*>(file) + pos, end - pos);
    if (!line.empty() && line.back() == '\r') line.pop_back();
    pos = end + 1;
    return true;
}

}  // namespace

HdrImage ParseRadianceHDR(const uint8_t* file, size_t bytes) {
    if (!file || bytes < 11) throw HipException("hdr: file too small");
    size_t pos = 0;
    std::string line;
    if (!NextLine(file, bytes, pos, line) || (line != "#?RADIANCE" && line != "#?RGBE")) throw HipException("hdr: missing #?RADIANCE signature");
    bool format_seen = false;
    for (;;) {
        if (!NextLine(file, bytes, pos, line)) throw HipException("hdr: header is not terminated by an empty line");
        if (line.empty()) break;
        if (line[0] == '#') continue;
        if (line.rfind("FORMAT=", 0) == 0) {
            if (line != "FORMAT=32-bit_rle_rgbe") throw HipException("hdr: unsupported " + line + " (only 32-bit_rle_rgbe)");
            format_seen = true;
        } else if (line.rfind("EXPOSURE=", 0) == 0) {
            if (std::strtof(line.c_str() + 9, nullptr) != 1.0f) throw HipException("hdr: EXPOSURE other than 1 is not supported");
        }
    }
    if (!format_seen) throw HipException("hdr: no FORMAT line");
    if (!NextLine(file, bytes, pos, line)) throw HipException("hdr: missing resolution line");
    unsigned h = 0, w = 0;
    char tail = 0;
    if (std::sscanf(line.c_str(), "-Y %u +X %u%c", &h, &w, &tail) != 2 || !w || !h || w > 32768 || h > 32768)
        throw HipException("hdr: unsupported resolution line '" + line + "' (only -Y h +X w)");

    HdrImage img;
    img.Width = w;
    img.Height = h;
    img.Rgbe.resize((size_t)w * h * 4);
    std::vector<uint8_t> planes((size_t)w * 4);
    for (unsigned y = 0; y < h; y++) {
        uint8_t* row = img.Rgbe.data() + (size_t)y * w * 4;
        const bool rle = w >= 8 && w < 32768 && pos + 4 <= bytes && file[pos] == 2 && file[pos + 1] == 2 && !(file[pos + 2] & 0x80);
        if (!rle) {   // flat scanline
            if (pos + (size_t)w * 4 > bytes) throw HipException("hdr: truncated flat scanline");
            std::memcpy(row, file + pos, (size_t)w * 4);
            pos += (size_t)w * 4;
            continue;
        }
        if ((unsigned)((file[pos + 2] << 8) | file[pos + 3]) != w) throw HipException("hdr: scanline length does not match the image width");
        pos += 4;
        for (int c = 0; c < 4; c++) {   // the four components are stored one after the other, each run-length coded
            uint8_t* dst = planes.data() + (size_t)c * w;
            unsigned x = 0;
            while (x < w) {
                if (pos >= bytes) throw HipException("hdr: truncated run-length data");
                unsigned count = file[pos++];
                if (count > 128) {   // run
                    count -= 128;
                    if (x + count > w || pos >= bytes) throw HipException("hdr: run overflows the scanline");
                    std::memset(dst + x, file[pos++], count);
                } else {             // literal
                    if (count == 0 || x + count > w || pos + count > bytes) throw HipException("hdr: bad literal count");
                    std::memcpy(dst + x, file + pos, count);
                    pos += count;
                }
                x += count;
            }
        }
        for (unsigned x = 0; x < w; x++) {
            row[4 * x + 0] = planes[x];
            row[4 * x + 1] = planes[(size_t)w + x];
            row[4 * x + 2] = planes[(size_t)2 * w + x];
            row[4 * x + 3] = planes[(size_t)3 * w + x];
        }
    }
    return img;
}

HdrImage LoadHDRImageFile(const std::string& path) {
    std::FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw HipException("hdr: cannot open " + path);
    std::vector<uint8_t> data;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    std::fclose(f);
    try {
        return ParseRadianceHDR(data.data(), data.size());
    } catch (const HipException& e) {
        throw HipException(path + ": " + e.what());
    }
}

std::shared_ptr<SkyBox> LoadCubeMap(pbr_ctx* ctx, const std::string& dir) {
    // same face order as the reference (ResourceLoader.cpp:415)
    static const char* const file_names[6] = {"px.hdr", "nx.hdr", "py.hdr", "ny.hdr", "pz.hdr", "nz.hdr"};
    HdrImage faces[6];
    for (int i = 0; i < 6; i++) {
        faces[i] = LoadHDRImageFile(dir + "/" + file_names[i]);
        if (faces[i].Width != faces[i].Height) throw HipException(std::string(file_names[i]) + ": cube faces must be square");
        if (faces[i].Width != faces[0].Width) throw HipException(std::string(file_names[i]) + ": cube faces differ in size");
        if (faces[i].Width % 4) throw HipException(std::string(file_names[i]) + ": width and height must be a multiple of 4 (ResourceLoader.cpp:399)");
    }
    const uint32 size = faces[0].Width;
    uint32 mips = 1;
    while ((size >> mips) >= 1) mips++;
    auto sky = std::make_shared<SkyBox>();
    sky->Cube = std::make_shared<DeviceTexture2DArray>(size, mips, ETextureFormat_R32G32B32A32_FLOAT);
    const size_t face_texels = (size_t)size * size;
    DeviceStructuredBuffer staging((uint32)(6 * face_texels * 4), 4);
    for (int i = 0; i < 6; i++)
        ThrowIfFailed(hipMemcpy((uint8_t*)staging.DevicePtr() + i * face_texels * 4, faces[i].Rgbe.data(), face_texels * 4, hipMemcpyHostToDevice), "upload rgbe face");
    auto check = [&](pbr_status st) { if (st != PBR_OK) throw HipException(pbr_last_error(ctx)); };
    check(pbr_rgbe_decode(ctx, (const uint8_t*)staging.DevicePtr(), 6 * face_texels, (float*)sky->Cube->DevicePtr()));
    check(pbr_cube_gen_mips(ctx, (float*)sky->Cube->DevicePtr(), size, mips));
    DeviceStructuredBuffer pack(112, 4);
    pbr_cube_f32 c{(const float*)sky->Cube->DevicePtr(), size, mips};
    check(pbr_sh9_project(ctx, &c, (float*)pack.DevicePtr()));
    check(pbr_sync(ctx));   // staging / pack are released on return
    ThrowIfFailed(hipMemcpy(&sky->SH, pack.DevicePtr(), 112, hipMemcpyDeviceToHost), "read SH");
    return sky;
}

}  // namespace MRendererHip
