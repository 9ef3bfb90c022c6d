/* Weight-layout contract of the drop-in boundary.
 *
 * These are the in-memory layer records that the reference's generated src/nnet_data.cpp
 * (written by dump_percepnet.py:56-155) instantiates and that its frame engine walks:
 * layer structs = reference src/nnet.h:44-95, model/state = reference src/nnet_data.h:6-38.
 * Field ORDER and TYPES are the ABI (the generated file uses positional initialisers); the
 * names are kept so a generated nnet_data.cpp compiles against this header unchanged.
 *
 * Array layouts (row-major C, float32):
 *   DenseLayer.input_weights   [nb_inputs][nb_neurons]
 *   Conv1DLayer.input_weights  [kernel_size][nb_inputs][nb_neurons]  (oldest tap first)
 *   GRULayer.input_weights     [nb_inputs ][3*nb_neurons]  gate order z, r, h
 *   GRULayer.recurrent_weights [nb_neurons][3*nb_neurons]  gate order z, r, h
 *   GRULayer.bias              [6*nb_neurons] = input z,r,h then recurrent z,r,h (reset_after)
 */
#ifndef PERCEPNET_NNET_DATA_H
#define PERCEPNET_NNET_DATA_H

#define ACTIVATION_LINEAR  0
#define ACTIVATION_SIGMOID 1
#define ACTIVATION_TANH    2
#define ACTIVATION_RELU    3
#define ACTIVATION_SOFTMAX 4

#define PN_CONV_DIM 512

typedef struct {
  const float *bias;
  const float *input_weights;
  int nb_inputs;
  int nb_neurons;
  int activation;
} DenseLayer;

typedef struct {
  const float *bias;
  const float *input_weights;
  const float *recurrent_weights;
  int nb_inputs;
  int nb_neurons;
  int activation;
  int reset_after;
} GRULayer;

typedef struct {
  const float *bias;
  const float *input_weights;
  int nb_inputs;
  int kernel_size;
  int nb_neurons;
  int activation;
} Conv1DLayer;

typedef struct RNNModel {
  const DenseLayer *fc;
  const Conv1DLayer *conv1;
  const Conv1DLayer *conv2;
  const GRULayer *gru1;
  const GRULayer *gru2;
  const GRULayer *gru3;
  const GRULayer *gru_gb;
  const GRULayer *gru_rb;
  const DenseLayer *fc_gb;
  const DenseLayer *fc_rb;
} RNNModel;

/* Per-stream network state exactly as the reference declares it (src/nnet_data.h:28-38): host arrays owned by the
   caller (rnnoise_init callocs them, denoise.cpp:268-274): conv FIFOs of kernel_size*nb_inputs floats of which the
   first (kernel_size-1)*nb_inputs are live (oldest frame first, nnet.cpp:191-199), GRU states of nb_neurons.
   compute_rnn(RNNState*, ...) (rnnoise.h:68, rnn.cpp:42) is exported by libpercepnet_hip with this record. */
typedef struct RNNState {
  const RNNModel *model;
  float *first_conv1d_state;
  float *second_conv1d_state;
  float *gru1_state;
  float *gru2_state;
  float *gru3_state;
  float *gb_gru_state;
  float *rb_gru_state;
  float convout_buf[PN_CONV_DIM * 3];
} RNNState;

#endif
