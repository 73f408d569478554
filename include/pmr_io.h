/* pmr_io.h -- ingest / egress formats around the accelerated chain.  Host-only C, no HIP.
 *
 *  pmr_iq_reader   stands in for SoapySDRDevice_readStream (reference src/sdr_pmr446.c:789, src/dsd_in.c:161; stream
 *                  format SOAPY_SDR_CF32, src/shared.c:62): recorded IQ from a file or stdin, delivered as cf32 blocks.
 *  pmr_wav_writer  stands in for the RtAudio sink (RTAUDIO_FORMAT_FLOAT32 at AUDIO_SAMPLERATE = 12.5 kHz, mono,
 *                  src/sdr_pmr446.c:585) and for dsd_in's stdout pipe (s16le mono 48 kHz, src/dsd_in.c:172-178,
 *                  README.md:45): a RIFF/WAVE file, or a headerless stream on stdout.
 * Conventions as in pmr_chain.h: opaque handles, open returns NULL on failure, int return codes, 0 == OK.
 */
#ifndef PMR_IO_H
#define PMR_IO_H

#include "pmr_chain.h"

#ifdef __cplusplus
extern "C" {
#endif

/* sample formats of a recording (SoapySDR names) */
enum { PMR_IQ_CF32 = 0,     /* interleaved float32 I/Q, little endian: what readStream delivers (src/shared.c:62)   */
       PMR_IQ_CS16 = 1,     /* interleaved int16 I/Q, scaled by 1/32768                                              */
       PMR_IQ_CU8 = 2 };    /* interleaved uint8 I/Q (rtl_sdr recordings), (x - 127.5) / 127.5                       */

typedef struct pmr_iq_reader_s *pmr_iq_reader;
pmr_iq_reader pmr_iq_reader_open(const char *path /* "-" = stdin */, int format);
/* Like readStream (:789): fills buf with up to max_samples samples and returns how many (short only at end of stream),
 * 0 at end of stream, a negative value on a read error.  A trailing partial sample is dropped.                     */
int  pmr_iq_reader_read(pmr_iq_reader r, pmr_cf32 *buf, unsigned max_samples);
int  pmr_iq_reader_close(pmr_iq_reader r);

enum { PMR_WAV_F32 = 0,     /* WAVE_FORMAT_IEEE_FLOAT, what the reference hands RtAudio (:585)                       */
       PMR_WAV_S16 = 1,     /* WAVE_FORMAT_PCM 16 bit                                                                 */
       PMR_RAW_S16 = 2 };   /* headerless s16le: the dsd_in wire format (src/dsd_in.c:177)                            */

typedef struct pmr_wav_writer_s *pmr_wav_writer;
pmr_wav_writer pmr_wav_writer_open(const char *path /* "-" = stdout (PMR_RAW_S16 only) */, int format,
                                   unsigned sample_rate, unsigned channels);
/* frames x channels samples; planar input like the chain's channel-major outputs: sample (c, t) at data[c * stride + t];
 * written interleaved.  The float variant takes float32 audio, the s16 variants take int16 PCM.                    */
int  pmr_wav_writer_write_f32(pmr_wav_writer w, const float *data, unsigned frames, unsigned stride);
int  pmr_wav_writer_write_s16(pmr_wav_writer w, const int16_t *data, unsigned frames, unsigned stride);
int  pmr_wav_writer_close(pmr_wav_writer w);      /* patches the RIFF sizes */

#ifdef __cplusplus
}
#endif
#endif
