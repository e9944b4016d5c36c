/* fake_rtlsdr.c -- TEST DOUBLE for librtlsdr (tests/test_host.py): the 13 entry
 * points RtlSdrSource resolves with dlsym, backed by a file replay instead of a
 * USB dongle.  Behaviour is driven by environment variables so that tests can
 * exercise the error paths:
 *   FAKE_RTLSDR_COUNT   number of "dongles" (default 1)
 *   FAKE_RTLSDR_FILE    u8 IQ bytes handed out sequentially by rtlsdr_read_sync
 *   FAKE_RTLSDR_LOG     every call is appended here as one text line
 *   FAKE_RTLSDR_RATE_OFFSET   actual rate = requested + offset (a dongle rounds rates)
 * Not part of the product; never shipped or linked. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint32_t freq, rate; int gain, ppm, gain_mode; FILE* data; } fake_dev;

static void say(const char* fmt, long a, long b)
{
    const char* path = getenv("FAKE_RTLSDR_LOG");
    if (!path) return;
    FILE* f = fopen(path, "a");
    if (!f) return;
    fprintf(f, fmt, a, b);
    fputc('\n', f);
    fclose(f);
}

uint32_t rtlsdr_get_device_count(void)
{
    const char* c = getenv("FAKE_RTLSDR_COUNT");
    return c ? (uint32_t)atoi(c) : 1u;
}

int rtlsdr_open(void** dev, uint32_t index)
{
    fake_dev* d = (fake_dev*)calloc(1, sizeof(fake_dev));
    const char* path = getenv("FAKE_RTLSDR_FILE");
    if (!d) return -1;
    d->data = path ? fopen(path, "rb") : NULL;
    *dev = d;
    say("open %ld", (long)index, 0);
    return 0;
}

int rtlsdr_close(void* dev)
{
    fake_dev* d = (fake_dev*)dev;
    if (d->data) fclose(d->data);
    free(d);
    say("close", 0, 0);
    return 0;
}

static const int kGains[] = {0, 9, 14, 27, 37, 77, 87, 125, 144, 157, 166, 197, 207, 229, 254, 280, 297,
                             328, 338, 364, 372, 386, 402, 421, 434, 439, 445, 480, 496};

int rtlsdr_get_tuner_gains(void* dev, int* gains)
{
    (void)dev;
    if (gains) memcpy(gains, kGains, sizeof kGains);
    return (int)(sizeof kGains / sizeof kGains[0]);
}

uint32_t rtlsdr_get_sample_rate(void* dev) { return ((fake_dev*)dev)->rate; }
uint32_t rtlsdr_get_center_freq(void* dev) { return ((fake_dev*)dev)->freq; }
int rtlsdr_reset_buffer(void* dev) { (void)dev; return 0; }

int rtlsdr_read_sync(void* dev, void* buf, int len, int* n_read)
{
    fake_dev* d = (fake_dev*)dev;
    *n_read = d->data ? (int)fread(buf, 1, (size_t)len, d->data) : 0;
    say("read %ld -> %ld", (long)len, (long)*n_read);
    return 0;
}

int rtlsdr_set_tuner_gain_mode(void* dev, int manual)
{
    ((fake_dev*)dev)->gain_mode = manual;
    say("gain_mode %ld", (long)manual, 0);
    return 0;
}

int rtlsdr_set_tuner_gain(void* dev, int gain)
{
    ((fake_dev*)dev)->gain = gain;
    say("gain %ld", (long)gain, 0);
    return 0;
}

int rtlsdr_set_center_freq(void* dev, uint32_t freq)
{
    ((fake_dev*)dev)->freq = freq;
    say("freq %ld", (long)freq, 0);
    return 0;
}

int rtlsdr_set_freq_correction(void* dev, int ppm)
{
    ((fake_dev*)dev)->ppm = ppm;
    say("ppm %ld", (long)ppm, 0);
    return 0;
}

int rtlsdr_set_sample_rate(void* dev, uint32_t rate)
{
    const char* off = getenv("FAKE_RTLSDR_RATE_OFFSET");
    ((fake_dev*)dev)->rate = rate + (off ? (uint32_t)atoi(off) : 0u);
    say("rate %ld", (long)rate, 0);
    return 0;
}
