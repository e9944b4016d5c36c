#include "options.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <map>
#include <sstream>
#include <vector>

#include "units.h"

namespace rpf_host {

const char* const kVersion = "1.0-beta2-mi355x";

namespace {

enum class Kind { Flag, Int, Int64, Double, Text };

struct Spec {
    char short_name;           // 0 = none
    const char* long_name;
    Kind kind;
    const char* value_name;
    const char* help;
};

// Same options, names and help texts as params.cxx:104-141 (alphabetical like the man page).
const Spec kSpecs[] = {
    {'B', "baseline", Kind::Text, "file|-", "Subtract baseline, read baseline data from file or stdin."},
    {'b', "bins", Kind::Int, "bins in FFT spectrum", "Number of bins in FFT spectrum (must be even number)"},
    {0, "buffers", Kind::Int, "buffers", "Number of read buffers (don't touch unless running out of memory)."},
    {'c', "continue", Kind::Flag, "", "Repeat the same measurement endlessly."},
    {'d', "device", Kind::Int, "device index", "RTL-SDR device index."},
    {'e', "elapsed", Kind::Text, "seconds", "Scan session duration."},
    {'f', "freq", Kind::Text, "Hz | Hz:Hz", "Center frequency of the receiver or frequency range to scan."},
    {'g', "gain", Kind::Int, "1/10th of dB", "Receiver gain."},
    {'l', "linear", Kind::Flag, "", "Calculate linear power values instead of logarithmic."},
    {'m', "matrix", Kind::Text, "filename (without extension)",
     "Will output data in binary matrix format plus separate metadata text file"},
    {'n', "repeats", Kind::Int64, "repeats", "Number of scans for averaging (incompatible with -t)."},
    {'o', "overlap", Kind::Double, "percent",
     "Define lower boundary for overlap when frequency hopping (otherwise meaningless)."},
    {'p', "ppm", Kind::Int, "ppm", "Set custom ppm error in RTL-SDR device."},
    {'q', "quiet", Kind::Flag, "", "Limit verbosity."},
    {'r', "rate", Kind::Int, "samples/s", "Sample rate of the receiver."},
    {'s', "buffer-size", Kind::Int, "bytes", "Size of read buffers (leave it unless you know what you are doing)."},
    {'T', "strict-time", Kind::Flag, "",
     "End measurement when the time set with --time option is up, regardless of gathered samples."},
    {'t', "time", Kind::Text, "seconds", "Integration time (incompatible with -n)."},
    {'w', "window", Kind::Text, "file|-", "Use window function, from file or stdin."},
    // additive (not in the reference)
    {0, "input", Kind::Text, "file|-", "Replay interleaved 8-bit IQ samples from a file or stdin."},
    {0, "synthetic", Kind::Int64, "seed", "Use the built-in synthetic receiver instead of a dongle."},
    {0, "gpu", Kind::Int, "ordinal", "HIP device to run on."},
    {0, "gpus", Kind::Text, "a,b,...", "HIP devices to spread a scan over (one engine per listed device)."},
    {0, "reduce", Kind::Text, "rccl|host", "With --gpus: where a scan's per-device spectra are added (default: rccl if it loads, else host)."},
    {'h', "help", Kind::Flag, "", "Displays usage information and exits."},
    {0, "version", Kind::Flag, "", "Displays version information and exits."},
};

struct Parsed {
    std::map<std::string, std::string> value;   // long name -> text ("" for flags)
    bool has(const char* n) const { return value.count(n) != 0; }
    const std::string& get(const char* n) const { return value.at(n); }
};

[[noreturn]] void parse_error(const std::string& what, const std::string& arg)
{
    // TCLAP's ArgException text as rethrown at params.cxx:266-270
    throw RPFexception("Error: " + what + " for arg " + arg, ReturnValue::TCLAPerror);
}

const Spec* find_spec(const std::string& token)
{
    for (const Spec& s : kSpecs) {
        if (token.size() == 2 && token[0] == '-' && s.short_name && token[1] == s.short_name) return &s;
        if (token.size() > 2 && token.compare(0, 2, "--") == 0 && token.substr(2) == s.long_name) return &s;
    }
    return nullptr;
}

template <typename T>
T to_number(const Spec& s, const std::string& text)
{
    std::istringstream in(text);
    T v{};
    if (!(in >> v) || !(in >> std::ws).eof())
        parse_error("Couldn't read argument value from string '" + text + "'",
                    std::string("--") + s.long_name);
    return v;
}

Parsed tokenize(int argc, const char* const* argv)
{
    Parsed out;
    for (int i = 1; i < argc; ++i) {
        std::string token = argv[i];
        if (token == "--" || token == "--ignore_rest") break;
        std::string attached;
        bool has_attached = false;
        // --name=value
        const size_t eq = token.find('=');
        if (token.compare(0, 2, "--") == 0 && eq != std::string::npos) {
            attached = token.substr(eq + 1);
            token = token.substr(0, eq);
            has_attached = true;
        }
        const Spec* spec = find_spec(token);
        if (!spec) parse_error("Couldn't find match for argument", token);
        const std::string key = spec->long_name;
        if (out.has(key.c_str())) parse_error("Argument already set!", "--" + key);
        if (spec->kind == Kind::Flag) {
            out.value[key] = "";
            continue;
        }
        if (!has_attached) {
            if (i + 1 >= argc) parse_error("Missing a value for this argument!", "--" + key);
            attached = argv[++i];
        }
        out.value[key] = attached;
    }
    return out;
}

void require_non_negative(const Parsed& p, const char* name)
{
    if (!p.has(name)) return;
    std::istringstream in(p.get(name));
    double v = 0;
    in >> v;
    if (v < 0)
        throw RPFexception(std::string("Argument to '") + name + "' must be a positive number.",
                           ReturnValue::InvalidArgument);
}

}  // namespace

std::string usage_text()
{
    std::ostringstream o;
    o << "USAGE:\n   rpf_power [OPTION ...]\n\nObtain power spectrum from RTL device using FFTW library"
         " (MI355X engine build).\n\nWhere:\n";
    for (const Spec& s : kSpecs) {
        o << "   ";
        if (s.short_name) o << "-" << s.short_name << ",  ";
        o << "--" << s.long_name;
        if (s.kind != Kind::Flag) o << " <" << s.value_name << ">";
        o << "\n     " << s.help << "\n\n";
    }
    return o.str();
}

Options parse_command_line(int argc, const char* const* argv)
{
    const Parsed p = tokenize(argc, argv);
    Options o;
    if (p.has("help")) { o.show_help = true; return o; }
    if (p.has("version")) { o.show_version = true; return o; }

    // validate the types first (TCLAP does this while parsing)
    for (const Spec& s : kSpecs) {
        if (!p.has(s.long_name)) continue;
        if (s.kind == Kind::Int) (void)to_number<int>(s, p.get(s.long_name));
        if (s.kind == Kind::Int64) (void)to_number<int64_t>(s, p.get(s.long_name));
        if (s.kind == Kind::Double) (void)to_number<double>(s, p.get(s.long_name));
    }
    // params.cxx:146-147
    for (const char* name : {"bins", "rate", "gain", "device", "buffers", "buffer-size", "repeats"})
        require_non_negative(p, name);

    auto int_of = [&](const char* n, int fallback) {
        return p.has(n) ? to_number<int>(*find_spec(std::string("--") + n), p.get(n)) : fallback;
    };
    o.dev_index = int_of("device", o.dev_index);
    o.N = int_of("bins", o.N);
    if (o.N % 2 != 0) {                                              // params.cxx:150-155
        o.N++;
        std::cerr << "Number of bins should be even, changing to " << o.N << "." << std::endl;
    }
    o.linear = p.has("linear");
    o.gain = int_of("gain", o.gain);
    o.sample_rate = int_of("rate", o.sample_rate);
    o.buffers = int_of("buffers", o.buffers);
    o.buf_length = int_of("buffer-size", o.buf_length);
    o.endless = p.has("continue");
    o.talkless = p.has("quiet");
    o.strict_time = p.has("strict-time");
    if (p.has("overlap")) o.min_overlap = to_number<double>(*find_spec("--overlap"), p.get("overlap"));
    o.device = int_of("gpu", 0);
    if (p.has("gpus")) {
        if (p.has("gpu"))
            throw RPFexception("Options --gpu and --gpus are mutually exclusive. Exiting.", ReturnValue::InvalidArgument);
        const std::string list = p.get("gpus");
        size_t pos = 0;
        while (pos <= list.size()) {
            const size_t comma = std::min(list.find(',', pos), list.size());
            const std::string item = list.substr(pos, comma - pos);
            char* end = nullptr;
            const long v = std::strtol(item.c_str(), &end, 10);
            if (item.empty() || *end != '\0' || v < 0 || v > 4095)
                throw RPFexception("Could not parse the device list given to --gpus: " + list + ".\n"
                                   "Expecting comma-separated HIP device ordinals. Exiting.",
                                   ReturnValue::InvalidArgument);
            o.devices.push_back(static_cast<int>(v));
            pos = comma + 1;
        }
        o.device = o.devices.front();
    } else {
        o.devices.push_back(o.device);
    }
    if (p.has("reduce")) {
        o.reduce = p.get("reduce");
        if (o.reduce != "rccl" && o.reduce != "host")
            throw RPFexception("Argument to --reduce must be rccl or host. Exiting.", ReturnValue::InvalidArgument);
    }

    if (o.buf_length % base_buf != 0) {                              // params.cxx:171-175
        o.buf_length = static_cast<int>(std::floor(static_cast<double>(o.buf_length) / base_buf + 0.5) * base_buf);
        std::cerr << "Buffer length should be multiple of " << base_buf << ", changing to "
                  << o.buf_length << "." << std::endl;
    }
    o.ppm_error = int_of("ppm", o.ppm_error);

    if (p.has("freq")) {                                             // params.cxx:177-212
        const std::string text = p.get("freq");
        const size_t colon = text.find(':');
        if (colon != std::string::npos) {
            const std::string lo = text.substr(0, colon), hi = text.substr(colon + 1);
            if (lo.empty() || hi.empty())
                throw RPFexception("Could not parse frequency range given to --freq: " + text + ".\n"
                                   "Expecting form startfreq:stopfreq. Exiting.", ReturnValue::InvalidArgument);
            o.startfreq = parse_frequency(lo);
            o.stopfreq = parse_frequency(hi);
            if (o.startfreq < 0 || o.stopfreq < 0 || o.stopfreq < o.startfreq)
                throw RPFexception("Invalid frequency range given to --freq: " + text + ".\n"
                                   "Expecting positive numbers in ascending order, allowing the k,M,G "
                                   "multipliers. Exiting.", ReturnValue::InvalidArgument);
            o.freq_hopping_isSet = true;
            o.cfreq = (o.startfreq + o.stopfreq) / 2;
        } else {
            o.cfreq = parse_frequency(text);
            if (o.cfreq < 0)
                throw RPFexception("Invalid frequency given to --freq: " + std::to_string(o.cfreq) + ".\n"
                                   "Expecting a positive number, allowing the k,M,G multipliers. Exiting.",
                                   ReturnValue::InvalidArgument);
        }
    }

    if (p.has("repeats")) o.repeats = to_number<int64_t>(*find_spec("--repeats"), p.get("repeats"));
    else o.repeats = o.buf_length / (2 * o.N);                        // params.cxx:214-217
    if (p.has("time")) {
        o.integration_time = parse_time(p.get("time"));
        if (o.integration_time <= 0)
            throw RPFexception("Could not parse the value given to --time. Expecting format [WdXhYm]Z[s]. Exiting.",
                               ReturnValue::InvalidArgument);
        o.integration_time_isSet = true;
    }
    if (p.has("time") && p.has("repeats"))
        throw RPFexception("Options -n and -t are mutually exclusive. Exiting.", ReturnValue::InvalidArgument);
    if (o.strict_time && !p.has("time")) {
        std::cerr << "Warning: option --strict-time has no effect without --time." << std::endl;
        o.strict_time = false;
    }
    o.buf_length_isSet = p.has("buffer-size");
    o.baseline = p.has("baseline");
    if (o.baseline) o.baseline_file = p.get("baseline");
    o.window = p.has("window");
    if (o.window) o.window_file = p.get("window");
    o.matrixMode = p.has("matrix");
    if (o.matrixMode) {
        o.matrix_file = p.get("matrix");
        o.bin_file = o.matrix_file + ".bin";
        o.meta_file = o.matrix_file + ".met";
    }
    if (p.has("elapsed")) {
        o.session_duration = parse_time(p.get("elapsed"));
        if (o.session_duration <= 0)
            throw RPFexception("Could not parse the value given to --time. Expecting format [WdXhYm]Z[s]. Exiting.",
                               ReturnValue::InvalidArgument);
        o.session_duration_isSet = true;
    }
    if (p.has("input")) o.input_file = p.get("input");
    o.synthetic = p.has("synthetic");
    if (p.has("synthetic")) o.synthetic_seed = static_cast<uint64_t>(to_number<int64_t>(*find_spec("--synthetic"), p.get("synthetic")));
    return o;
}

}  // namespace rpf_host
