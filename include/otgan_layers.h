/* otgan_layers.h -- layer kernels of the generator / critic (included from otgan.h). */
#ifndef OTGAN_LAYERS_H
#define OTGAN_LAYERS_H
#endif
