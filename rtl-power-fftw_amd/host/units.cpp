#include "units.h"

#include <sstream>

namespace rpf_host {

int64_t parse_frequency(const std::string& text)
{
    std::istringstream in(text);
    double value = 0;
    std::string suffix;
    in >> value >> suffix;
    double scale = 1;
    if (suffix == "k") scale = 1e3;
    else if (suffix == "M") scale = 1e6;
    else if (suffix == "G") scale = 1e9;
    else if (!suffix.empty()) return -1;
    return static_cast<int64_t>(value * scale);
}

double parse_time(const std::string& text)
{
    std::string s = text;
    if (s.empty()) return -1;
    // a bare trailing number counts as seconds
    if (std::string("dhms").find(s.back()) == std::string::npos) s.push_back('s');

    static const struct { char unit; double seconds; } kUnits[] = {
        {'d', 86400.0}, {'h', 3600.0}, {'m', 60.0}, {'s', 1.0}};
    bool used[4] = {false, false, false, false};

    std::stringstream in(s);
    double total = 0, value = 0;
    char unit = 0;
    while (in >> value && in.get(unit)) {
        int which = -1;
        for (int i = 0; i < 4; ++i)
            if (kUnits[i].unit == unit) which = i;
        if (which < 0 || used[which]) return -1;     // unknown or repeated unit
        used[which] = true;
        total += value * kUnits[which].seconds;
    }
    return in.eof() ? total : -1;                    // leftovers mean a parse error
}

}  // namespace rpf_host
