// HdrImage.h — Radiance .hdr (RGBE) file ingestion for sky cubes.
//
// Reference: ResourceLoader::LoadHDRImageFile / LoadCubeMap (Engine/Source/Resource/ResourceLoader.cpp:381-428)
// hand the six faces px/nx/py/ny/pz/nz.hdr to DirectX::LoadFromHDRFile and then to GenerateMipMaps
// (:465-507).  DirectXTex is an un-vendored vcpkg dependency; the parser below follows the published file
// format (Radiance "picture" files: text header, "-Y h +X w" resolution line, flat or new-style
// run-length-encoded RGBE scanlines).  Split of work: the byte-serial parse and RLE expansion stay on the
// host (a face is <= a few MB), the RGBE -> fp32 conversion and the mip chain run on the GPU
// (pbr_rgbe_decode, pbr_cube_gen_mips).
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "Scene.h"

namespace MRendererHip {

struct HdrImage {
    uint32_t Width = 0, Height = 0;
    std::vector<uint8_t> Rgbe;   // Width * Height * 4, rows top to bottom
};

// Parses a whole .hdr file held in memory.  Throws HipException with a reason on malformed input
// (bad magic, unsupported FORMAT / orientation, truncated or inconsistent scanline data).
HdrImage ParseRadianceHDR(const uint8_t* file, size_t bytes);
HdrImage LoadHDRImageFile(const std::string& path);

// LoadCubeMap (ResourceLoader.cpp:408-428): <dir>/{px,nx,py,ny,pz,nz}.hdr -> fp32 RGBA cube with the full box
// mip chain and its SH9 pack.  Faces must be square, equal, and a multiple of 4 texels (:399-403).
std::shared_ptr<SkyBox> LoadCubeMap(pbr_ctx* ctx, const std::string& dir);

}  // namespace MRendererHip
