#include "sample_source.h"

#include <algorithm>

namespace rpf_host {

FileSource::FileSource(const std::string& path)
{
    if (path == "-") {
        file_ = stdin;
    } else {
        file_ = std::fopen(path.c_str(), "rb");
        owns_ = true;
        if (!file_) throw RPFexception("Could not open " + path + ". Quitting.", ReturnValue::InvalidInput);
    }
}

FileSource::~FileSource()
{
    if (file_ && owns_) std::fclose(file_);
}

bool FileSource::read(Buffer& buffer)
{
    const size_t got = std::fread(buffer.data(), 1, buffer.size(), file_);
    return got == buffer.size();
}

namespace {

inline uint64_t splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

inline int floor_div_256(int v) { return v >= 0 ? v / 256 : -((-v + 255) / 256); }

const int kTone8C[8] = {10, 7, 0, -7, -10, -7, 0, 7};
const int kTone8S[8] = {0, 7, 10, 7, 0, -7, -10, -7};
const int kTone16C[16] = {6, 6, 4, 2, 0, -2, -4, -6, -6, -6, -4, -2, 0, 2, 4, 6};
const int kTone16S[16] = {0, 2, 4, 6, 6, 6, 4, 2, 0, -2, -4, -6, -6, -6, -4, -2};

inline uint8_t clip_u8(int v) { return static_cast<uint8_t>(std::min(255, std::max(0, v))); }

}  // namespace

void SyntheticSource::generate(uint64_t seed, uint64_t first, uint64_t n, uint8_t* out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t k = first + i;
        const uint64_t w = splitmix64_at(seed, k);
        int si = 0, sq = 0;
        for (int b = 0; b < 4; ++b) {
            si += static_cast<int>((w >> (8 * b)) & 0xff);
            sq += static_cast<int>((w >> (8 * (b + 4))) & 0xff);
        }
        const int ni = floor_div_256((si - 510) * 35), nq = floor_div_256((sq - 510) * 35);
        const int ti = kTone8C[k % 8] + kTone16C[(3 * k) % 16];
        const int tq = kTone8S[k % 8] + kTone16S[(3 * k) % 16];
        out[2 * i] = clip_u8(128 + ni + ti);
        out[2 * i + 1] = clip_u8(128 + nq + tq);
    }
}

void SyntheticSource::set_frequency(int64_t hz)
{
    frequency_ = hz;
    seed_ = seed_for(base_seed_, hz);
    position_ = 0;
}

bool SyntheticSource::read(Buffer& buffer)
{
    const uint64_t n = buffer.size() / 2;
    generate(seed_, position_, n, buffer.data());
    position_ += n;
    return true;
}

}  // namespace rpf_host
