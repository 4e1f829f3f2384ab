"""Hot-path packages of the MI355X drop-in: renderer, scene, network, signal_encoder."""
